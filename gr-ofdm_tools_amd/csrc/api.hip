// C ABI of libofdmtools_hip.so (see include/ofdm_tools_hip.h).  Host side only:
// argument checking, plan bookkeeping, device scratch, launch ordering.  All
// arithmetic of the path runs in the kernels of this directory; there is no CPU
// fallback.
#include "../../include/ofdm_tools_hip.h"
#include "oth_internal.h"
#include "abi_barrier.h"

#include <sched.h>
#include <time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace oth;

struct oth_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;
    std::string err;
    std::string name;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    double total_ms = 0.0;
    uint64_t launches = 0;
    std::map<int, float2 *> twiddles;
    float *sink = nullptr;
    double *acc4 = nullptr;
    unsigned *queue = nullptr;         // 64 chunk tickets for the dynamic segment schedule
    unsigned *done_count = nullptr;    // arrival counter of a finalize launch that signals a polling host (FinalizeArgs)
    std::recursive_mutex mu;           // every entry point that takes this context (or a plan / chain of it) holds it
    bool queue_clean = false;          // all zero on the stream's timeline (finalize_kernel re-zeroes what a launch used)
    int queue_used = 0;                // counters the last averaging launch drew from
    unsigned char *scratch = nullptr;  // device scratch of the small ops (channel power, decision stage, xcorr): grown on
    size_t scratch_cap = 0;            // demand, never freed per call
    // channel slice bounds of the decision stage (oth_scan_decide_dev*): a scanner passes the same lo / hi on every
    // call, so they live on the device and are uploaded again only when their contents change - through a pinned
    // staging buffer, so that the upload is a real asynchronous copy (from pageable memory hipMemcpyAsync may hold the
    // host until the stream has drained, which would make the "asynchronous" entry points wait for the PSD kernels)
    std::vector<int> bounds_host;      // lo[nch] then hi[nch], as last uploaded
    int *d_bounds = nullptr;
    int *h_bounds = nullptr;           // pinned
    size_t bounds_cap = 0;             // ints
    hipEvent_t bounds_ev = nullptr;    // behind the last upload: the pinned words may be rewritten after it
};

// Device tables and scratch of the any-length route (fft_any.hip) of one plan / chain
struct AnyTables {
    AnyShape sh{};                     // kind ANY_NONE: not in use
    const float2 *tw = nullptr;        // W_L^k, L entries (the context's table cache owns it)
    float2 *chirp = nullptr;           // Bluestein c[n] = exp(-i pi n^2 / nfft), nfft entries
    float2 *midtab = nullptr;          // Bluestein FFT_M(conj c) / M, M entries by natural index
    float2 *ws = nullptr;              // workspace [channel][segment of the chunk][L]
    size_t ws_cap = 0;
    float4 *mean = nullptr;            // [channel][segment of the chunk] hi / lo means
    size_t mean_cap = 0;
};

struct oth_plan {
    oth_ctx *ctx = nullptr;
    int nfft = 0, nperseg = 0, noverlap = 0, step = 0, detrend = 0, scaling = 0, fftshift = 0, trim = 0;
    int db = 0, kernel = OTH_KERNEL_AUTO, sched = OTH_SCHED_DYNAMIC;
    double fs = 1.0, scale = 1.0;      // scale applies to the MEAN over segments
    float *d_win = nullptr;
    const float2 *d_tw = nullptr;
    float4 *d_fd = nullptr;            // window spectrum for the frequency-domain detrend (welch4096ws), or nullptr
    float4 *d_fd1x = nullptr;          // the same for welch16k1x_half_kernel (16384 points, spectrum confined to |k| < 16)
    float2 *d_pilot = nullptr;         // per-stream pilots of the frequency-domain detrend (WelchArgs.pilot)
    size_t pilot_cap = 0;
    bool fast_detrend = false;         // OTH_DETREND_CONSTANT_FAST: the builds without the pilot (WelchArgs.pilot)
    bool rect_window = false;          // every window value is 1 (window == NULL or boxcar): builds without the multiply
    float *d_partial = nullptr;
    size_t partial_cap = 0;
    int last_W = 0;
    float *d_reduce = nullptr;         // stage-1 output of the two-stage partial-sum reduction
    size_t reduce_cap = 0;
    float *d_out = nullptr;            // [4][nfft] + pxy extra
    size_t out_cap = 0;
    // Host-output ring of oth_welch_exec / _exec_async (round 5).  The finalize launch writes the PSD straight into a
    // pinned, device-visible row (no copy-engine hop) and then a completion word next to it (FinalizeArgs.host_seq);
    // the host polls that word instead of sleeping in hipStreamSynchronize.  A slot is reused kOutRing launches later.
    static constexpr int kOutRing = 4;
    float *h_out = nullptr;            // pinned [kOutRing][nfft]
    unsigned *h_seq = nullptr;         // pinned [kOutRing]: low 32 bits of the ticket whose row is complete
    uint64_t out_ticket[kOutRing] = {0, 0, 0, 0};
    uint64_t out_nseg[kOutRing] = {0, 0, 0, 0};
    uint64_t next_out_ticket = 1;
    bool pilot_launch = false;         // A/B + parity: the pilot from pilot_mean_kernel also where the kernel could form it
    std::string last_recipe;           // recipe_text() of the last averaging launch (oth__debug_last_recipe)
    float2 *d_stage = nullptr;         // host-input staging (x then y)
    size_t stage_cap = 0;
    // streaming state
    float *d_sum = nullptr;            // raw sum |X|^2, natural order
    float *d_wpm = nullptr;            // 65536-point plans on welch32k.hip: w[n] + w[n + 32768], then w[n] - w[n + 32768] (n < 32768)
    uint64_t nseg_total = 0;
    size_t carry = 0;                  // samples kept at the front of d_stream
    float2 *d_stream = nullptr;
    size_t stream_cap = 0;
    // launch tuning (A/B tools and the parity suite): the OTH_W4096_* environment variables are read ONCE, when
    // the plan is created; oth_plan_set_tuning() changes them afterwards.  0 / -1 / empty = library default.
    std::string tune_variant;
    int tune_sched = -1, tune_chunk = 0, tune_tail = 0;
    // pinned staging ring of the streaming form (oth_welch_accumulate): the caller's buffer is copied here, the
    // H2D copy and the kernels are enqueued, and the call returns without waiting for the GPU
    void *h_ring[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t h_ring_cap[4] = {0, 0, 0, 0};
    hipEvent_t h_ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned h_ring_next = 0;
    AnyTables any;                     // any.sh.kind != ANY_NONE: the plan's length runs through fft_any.hip
    // Blocking oth_welch_exec calls on one plan run one at a time (enqueue + collect under this mutex; the CONTEXT lock is
    // free while they wait): with the 4-slot output ring a fifth concurrent caller's launch would otherwise rewrite the
    // first caller's row before it was copied out (advisor, round 5).
    std::mutex exec_mu;
    int hostwait = 0;                  // 0 poll the completion word (default), 1 hipStreamSynchronize (oth_plan_set_hostwait)
};

struct oth_chain {
    oth_ctx *ctx = nullptr;
    int nfft = 0, fftshift = 0, epilogue = 0, keep_n = 1, count = 1;
    float *d_win = nullptr;
    const float2 *d_tw = nullptr;
    float2 *d_buf = nullptr;           // leftover + new samples
    size_t buf_cap = 0;
    size_t leftover = 0;               // samples at the front of d_buf
    float *d_rows = nullptr;
    size_t rows_cap = 0;
    int do_iir = 0, do_peak = 0;
    float alpha = 0.f, kdb = 0.f;
    float *d_iir = nullptr, *d_peak = nullptr;
    int *d_peak_init = nullptr;
    float2 *d_stage = nullptr;         // host input lands here (H2D), then feeds the kernels
    size_t stage_cap = 0;
    float *d_partial = nullptr;        // per-team accumulator rows of the fused kernel
    size_t partial_cap = 0;
    float *d_tail = nullptr;           // group rows of the two-launch cross-team reduction
    size_t tail_cap = 0;
    bool peak_flag_set = false;        // d_peak_init is 1 on the stream's timeline
    bool rect = false;                 // the window is all ones (fft_vcc's `()`): the 8192 / 16384 chain skips the multiply
    int kernel = OTH_KERNEL_AUTO;      // OTH_KERNEL_GENERIC forces the coverage kernels (parity tests)
    float *d_out = nullptr;            // rows handed back by the host-output forms
    size_t out_cap = 0;
    // asynchronous work() form (oth_chain_push_async): pinned input ring + pinned latest-row ring.  A slot is
    // reused kRing pushes later; the push waits only if the GPU is still that far behind.
    static constexpr int kRing = 4;
    void *h_in[kRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t h_in_cap[kRing] = {0, 0, 0, 0};
    float *h_row[kRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kRing] = {nullptr, nullptr, nullptr, nullptr};      // recorded behind the D2H of the slot's row
    uint64_t ticket_of[kRing] = {0, 0, 0, 0};
    uint64_t nrows_of[kRing] = {0, 0, 0, 0};
    uint64_t next_ticket = 1;
    AnyTables any;                     // any.sh.kind != ANY_NONE: the chain's length runs through fft_any.hip
    // round 6: what a work()-sized push costs
    uint64_t ops = 0;                  // stream operations (asynchronous copies + kernel launches) the last push enqueued
    bool noop[kRing] = {false, false, false, false};      // the slot's push enqueued nothing (every vector dropped): ready at once
    bool leftover_stale = false;       // the partial vector in d_buf was not copied (it is not a kept one): never emit it
};

namespace {

// host chunks up to this size go through the pinned staging rings (work()-sized buffers: the copy is trivial and the
// call returns at once); larger ones use the runtime's staged copy from pageable memory directly
constexpr size_t kPinnedStageMax = 1u << 20;
// a PINNED / registered source above this size is not copied into the ring (that would pin as much again): its DMA is
// enqueued directly and the call waits for that one copy - the only case in which a push waits for the stream
constexpr size_t kPinnedRingMax = 64u << 20;

thread_local std::string g_err = "no error";

// fewest segments per stream for which a detrending plan picks the frequency-domain detrend builds (run_average)
constexpr long long kFdMinSegments = 8;

// A host buffer the runtime can DMA from directly (hipHostMalloc / hipHostRegister'd, e.g. a torch pinned tensor or a
// registered scheduler buffer): hipMemcpyAsync from it returns before the bytes are read, so the "input valid only
// during the call" contract of work() needs a copy that has finished when the call returns.  Pageable memory is
// staged by the runtime before hipMemcpyAsync returns.
bool host_ptr_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // unregistered pageable memory: an error on older runtimes, not sticky
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

int fail(oth_ctx *c, int code, const std::string &msg) {
    if (c)
        c->err = msg;
    else
        g_err = msg;
    return code;
}

// Serialises the entry points per context: GNU Radio runs each block's work() on its own thread and the
// blocks of one process share the default context (scratch buffers, ticket counters, timing events).
struct CtxGuard {
    oth_ctx *c;
    explicit CtxGuard(oth_ctx *ctx) : c(ctx) {
        if (c) c->mu.lock();
    }
    ~CtxGuard() {
        if (c) c->mu.unlock();
    }
    CtxGuard(const CtxGuard &) = delete;
    CtxGuard &operator=(const CtxGuard &) = delete;
};

#define HIPCHK(c, expr)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail((c), OTH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// The exception barrier of the C ABI (include/ofdm_tools_hip.h: "nothing throws or aborts").  Every extern "C" body
// sits between OTH_TRY and OTH_CATCH(context): a std::bad_alloc (std::vector / std::string growth), a
// std::system_error (the context's recursive mutex) or anything else a C++ runtime call may raise becomes an error
// code + last-error text instead of std::terminate() inside the host's ctypes call.  The handlers themselves must not
// throw: the text is stored through fail_nothrow().
int fail_nothrow(oth_ctx *c, int code, const char *what) noexcept {
    try {
        if (c)
            c->err = what;
        else
            g_err = what;
    } catch (...) {
    }
    return code;
}

// OTH_TRY / OTH_CATCH(context): csrc/abi_barrier.h (shared with the host-only probe the CPU suite builds)

int use_device(oth_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    return OTH_OK;
}

// H2D copy of a caller's host buffer that must be consumed before the call returns, without a ring slot
int copy_in_and_wait(oth_ctx *c, void *dst, const void *src, size_t bytes) {
    hipEvent_t ev = nullptr;
    HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipEventRecord(ev, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ev);
    hipEventDestroy(ev);
    if (e != hipSuccess) return fail(c, OTH_ERR_HIP, std::string("host copy: ") + hipGetErrorString(e));
    return OTH_OK;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int get_twiddles(oth_ctx *c, int nfft, const float2 **out) {
    auto it = c->twiddles.find(nfft);
    if (it != c->twiddles.end()) {
        *out = it->second;
        return OTH_OK;
    }
    std::vector<float2> h(nfft);
    for (int k = 0; k < nfft; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)nfft;
        h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    float2 *d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(float2) * nfft));
    HIPCHK(c, hipMemcpyAsync(d, h.data(), sizeof(float2) * nfft, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->twiddles[nfft] = d;
    *out = d;
    return OTH_OK;
}

template <typename P> int ensure(oth_ctx *c, P **ptr, size_t *cap, size_t need_bytes) {
    if (*cap >= need_bytes && *ptr) return OTH_OK;
    if (*ptr) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(*ptr));
        *ptr = nullptr;
        *cap = 0;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, need_bytes);
    if (e != hipSuccess) return fail(c, OTH_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    *ptr = reinterpret_cast<P *>(p);
    *cap = need_bytes;
    return OTH_OK;
}

// grow while keeping the first keep_bytes
template <typename P> int ensure_keep(oth_ctx *c, P **ptr, size_t *cap, size_t need_bytes, size_t keep_bytes) {
    if (*cap >= need_bytes && *ptr) return OTH_OK;
    void *p = nullptr;
    const size_t newcap = need_bytes + need_bytes / 4;
    hipError_t e = hipMalloc(&p, newcap);
    if (e != hipSuccess) return fail(c, OTH_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (*ptr) {
        if (keep_bytes) HIPCHK(c, hipMemcpyAsync(p, *ptr, keep_bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(*ptr));
    }
    *ptr = reinterpret_cast<P *>(p);
    *cap = newcap;
    return OTH_OK;
}

struct Timed {
    oth_ctx *c;
    hipEvent_t a = nullptr, b = nullptr;
    explicit Timed(oth_ctx *ctx) : c(ctx) {
        if (!c->timing) return;
        if (c->events.size() >= 8192) {   // fold what we have
            hipStreamSynchronize(c->stream);
            for (auto &ev : c->events) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) c->total_ms += ms;
                c->free_events.push_back(ev);
            }
            c->events.clear();
        }
        if (!c->free_events.empty()) {
            a = c->free_events.back().first;
            b = c->free_events.back().second;
            c->free_events.pop_back();
        } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            a = b = nullptr;
            return;
        }
        hipEventRecord(a, c->stream);
    }
    ~Timed() {
        if (!a) return;
        hipEventRecord(b, c->stream);
        c->events.emplace_back(a, b);
        c->launches++;
    }
};

// Build variants of the welch4096 kernel; OTH_W4096_VARIANT=<tag> selects one (experiments only).
struct W4096Variant {
    const char *tag;
    hipError_t (*launch)(const WelchArgs &, hipStream_t);
    int (*blocks_per_cu)();
    int chunk;      // default segments per chunk of the dynamic schedule (same-box A/B, tools/archive/ab_variants.py)
    int rows;       // rows of partial sums each workgroup writes
    bool fd;        // detrends in the frequency domain: needs WelchArgs.fd (a window with a confined spectrum)
    bool inline_pilot = false;      // forms the pilot of the constant detrend in its own prologue (WelchArgs.pilot_inline)
};
const W4096Variant kVariants[] = {
    {"dpp", launch_welch_tuned4096_dpp, tuned4096_blocks_per_cu_dpp, 8, 1, false},            // any step
    {"pipe", launch_welch_tuned4096_pipe, tuned4096_blocks_per_cu_pipe, 16, 1, false},        // step 2048 (50 % overlap)
    {"ws", launch_welch_tuned4096_ws, tuned4096_blocks_per_cu_ws, 20, 1, true, true},     // step 2048, confined window spectrum
#ifdef OTH_EXPERIMENTS
    {"ws2", launch_welch_tuned4096_ws2, tuned4096_blocks_per_cu_ws2, 20, 2, true},  // the same in one 1024-thread workgroup per CU (A/B, +7 %)
    {"diag", launch_welch_tuned4096_diag, tuned4096_blocks_per_cu_diag, 16, 1, false},      // stamped build (tools/archive/diag_stamps.py)
    {"exp1", launch_welch_tuned4096_exp1, tuned4096_blocks_per_cu_exp1, 16, 1, false},
    {"exp2", launch_welch_tuned4096_exp2, tuned4096_blocks_per_cu_exp2, 16, 1, false},
    {"exp3", launch_welch_tuned4096_exp3, tuned4096_blocks_per_cu_exp3, 16, 1, false},
    {"exp4", launch_welch_tuned4096_exp4, tuned4096_blocks_per_cu_exp4, 16, 1, false},
    {"wsx1", launch_welch_tuned4096_wsx1, tuned4096_blocks_per_cu_wsx1, 32, 1, true, true},
    {"wsx2", launch_welch_tuned4096_wsx2, tuned4096_blocks_per_cu_wsx2, 32, 1, true, true},
    {"wsx3", launch_welch_tuned4096_wsx3, tuned4096_blocks_per_cu_wsx3, 32, 1, true, true},
    {"wsx4", launch_welch_tuned4096_wsx4, tuned4096_blocks_per_cu_wsx4, 32, 1, true, true},
#endif
};
// the three shipped builds are looked up by tag, never by position: the table is edited between rounds
const W4096Variant *variant_by_tag(const char *tag) {
    for (const auto &v : kVariants)
        if (!strcmp(v.tag, tag)) return &v;
    return &kVariants[0];
}
const W4096Variant *w4096_variant(int step, bool fd_ok, const std::string &want) {
    const W4096Variant *const dpp = variant_by_tag("dpp"), *const pipe = variant_by_tag("pipe"), *const ws = variant_by_tag("ws");
    const W4096Variant *pick = (step == 2048) ? (fd_ok ? ws : pipe) : dpp;
    if (!want.empty())
        for (const auto &v : kVariants)
            if (want == v.tag) pick = &v;
    // the wave-specialised build detrends in the frequency domain: only with a confined window spectrum
    if (pick->fd && !fd_ok) pick = pipe;
    // the pipelined builds keep the overlapped half in registers: only for step = nperseg / 2
    if (pick != dpp && step != 2048) pick = dpp;
    return pick;
}

// Window spectrum table for welch4096ws (WelchArgs.fd).  FFT((x - m) w) = FFT(x w) - m FFT(w): the kernel
// corrects only bins [0, 256) and [3840, 4096), so the table exists only when FFT(w) is negligible elsewhere:
// |W[k]|^2 <= 1e-10 sum(w^2) there bounds the uncorrected term by 1e-10 |m|^2 / sigma^2 of a white-noise
// bin's level.  True for boxcar and the periodic cosine-sum windows (hann, blackman-harris, flattop, ...).
bool window_spectrum_table(const std::vector<float> &w, std::vector<float> &fd) {
    const int n = 4096;
    std::vector<double> re(n), im(n, 0.0);
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < 12; ++b) r |= ((i >> b) & 1) << (11 - b);
        re[r] = (double)w[i];
        s2 += (double)w[i] * (double)w[i];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < len / 2; ++j) {
                const double c = cos(ang * j), s = sin(ang * j);
                const int a = i + j, b = a + len / 2;
                const double tr = re[b] * c - im[b] * s, ti = re[b] * s + im[b] * c;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
    for (int k = 256; k < 3840; ++k)
        if (re[k] * re[k] + im[k] * im[k] > 1e-10 * s2) return false;
    fd.resize(4 * 256);
    for (int t = 0; t < 256; ++t) {
        const int k = (t >> 4) + 16 * (t & 15);
        fd[4 * t] = (float)re[k];
        fd[4 * t + 1] = (float)im[k];
        fd[4 * t + 2] = (float)re[k + 3840];
        fd[4 * t + 3] = (float)im[k + 3840];
    }
    return true;
}

// The same table for the role-split 1024 / 2048 kernel (segfft.hip, segws_kernel<R, 2>): consumer thread
// t = R k0 + j corrects, for k1 = j + R m (m < 16 / R), the bins k0 + 16 k1 (k2 = 0) and k0 + 16 k1 + 256 (R - 1).
bool window_spectrum_table_seg(const std::vector<float> &w, int n, std::vector<float> &fd) {
    const int R = n / 256, Q = 16 / R;
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    std::vector<double> re(n), im(n, 0.0);
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        re[r] = (double)w[i];
        s2 += (double)w[i] * (double)w[i];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < len / 2; ++j) {
                const double c = cos(ang * j), s = sin(ang * j);
                const int a = i + j, b = a + len / 2;
                const double tr = re[b] * c - im[b] * s, ti = re[b] * s + im[b] * c;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
    for (int k = 256; k < n - 256; ++k)
        if (re[k] * re[k] + im[k] * im[k] > 1e-10 * s2) return false;
    fd.assign((size_t)4 * (n / 16) * Q, 0.f);
    for (int t = 0; t < n / 16; ++t) {
        const int k0 = t / R, j = t % R;
        for (int m = 0; m < Q; ++m) {
            const int lo = k0 + 16 * (j + R * m), hi = lo + 256 * (R - 1);
            float *o = &fd[4 * ((size_t)Q * t + m)];
            o[0] = (float)re[lo];
            o[1] = (float)im[lo];
            o[2] = (float)re[hi];
            o[3] = (float)im[hi];
        }
    }
    return true;
}

// The same table for welch16k.hip (N = 4096 F, F = 2 or 4): thread tid = 256 k' + 16 k0 + k1 holds, after pass 3, the bins
// k' + F (k0 + 16 k1 + 256 k2); it corrects k2 = 0 and k2 = 15, i.e. the table exists when the window's spectrum is
// confined to [0, 256 F) U [N - 256 F, N).
bool window_spectrum_table_16k(const std::vector<float> &w, int n, std::vector<float> &fd) {
    const int F = n / 4096;
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    std::vector<double> re(n), im(n, 0.0);
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        re[r] = (double)w[i];
        s2 += (double)w[i] * (double)w[i];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < len / 2; ++j) {
                const double c = cos(ang * j), s = sin(ang * j);
                const int a = i + j, b = a + len / 2;
                const double tr = re[b] * c - im[b] * s, ti = re[b] * s + im[b] * c;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
    for (int k = 256 * F; k < n - 256 * F; ++k)
        if (re[k] * re[k] + im[k] * im[k] > 1e-10 * s2) return false;
    fd.assign((size_t)4 * 256 * F, 0.f);
    for (int tid = 0; tid < 256 * F; ++tid) {
        const int kp = tid >> 8, k0 = (tid >> 4) & 15, k1 = tid & 15;
        const int lo = kp + F * (k0 + 16 * k1), hi = kp + F * (k0 + 16 * k1 + 3840);
        fd[4 * tid] = (float)re[lo];
        fd[4 * tid + 1] = (float)im[lo];
        fd[4 * tid + 2] = (float)re[hi];
        fd[4 * tid + 3] = (float)im[hi];
    }
    return true;
}

// The table for welch16k1x_half_kernel.  N = 16384: thread tid = 64 k0 + 4 k1 + q corrects register k2 = 0 - bin k0 when
// k1 = 0, q = 0 - and register k2 = 15 - bin N - 16 + k0 when k1 = 15, q = 3, where the quad butterfly leaves i X.
// N = 8192 (the 8-wave form, round 5): wave k0' holds k0 = 2 k0' + h; register k2 = 0 of lane 32 h is bin k0, register
// k2 = 15 of lane 32 h + 31 (k1 = 7, q = 3) is bin N - 16 + k0.  Either way the table exists when the window's spectrum is
// confined to |k| < 16 (all periodic cosine-sum windows; boxcar).
bool window_spectrum_table_1x(const std::vector<float> &w, int n, std::vector<float> &fd) {
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    std::vector<double> re(n), im(n, 0.0);
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        re[r] = (double)w[i];
        s2 += (double)w[i] * (double)w[i];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int j = 0; j < len / 2; ++j) {
                const double c = cos(ang * j), s = sin(ang * j);
                const int a = i + j, b = a + len / 2;
                const double tr = re[b] * c - im[b] * s, ti = re[b] * s + im[b] * c;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
    for (int k = 16; k < n - 16; ++k)
        if (re[k] * re[k] + im[k] * im[k] > 1e-10 * s2) return false;
    fd.assign((size_t)4 * (n / 16), 0.f);
    for (int k0 = 0; k0 < 16; ++k0) {
        const int lo = n == 16384 ? 64 * k0 : 64 * (k0 >> 1) + 32 * (k0 & 1);
        const int hi = n == 16384 ? 64 * k0 + 63 : 64 * (k0 >> 1) + 32 * (k0 & 1) + 31;
        fd[4 * lo] = (float)re[k0];
        fd[4 * lo + 1] = (float)im[k0];
        const int kh = n - 16 + k0;                       // i (re + i im) = -im + i re
        fd[4 * hi + 2] = (float)(-im[kh]);
        fd[4 * hi + 3] = (float)re[kh];
    }
    return true;
}

int segments(const oth_plan *p, size_t nsamples, long long *nseg) {
    if (nsamples < (size_t)p->nperseg) return OTH_ERR_INVALID;
    *nseg = (long long)((nsamples - (size_t)p->noverlap) / (size_t)p->step);
    return OTH_OK;
}

// ---- the any-length route (fft_any.hip, round 6) -----------------------------------------------------------------------
// Lengths the power-of-two kernels do not take: any_describe() picks direct / two-level / Bluestein; the tables below are
// built once per plan or chain, any_run() drives the launches of one averaging (or periodogram-row) request.
void host_fft_pow2(std::vector<double> &re, std::vector<double> &im) {      // in place, forward, n a power of two
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t j = 0; j < len / 2; ++j) {
            const double c = cos(ang * (double)j), sn = sin(ang * (double)j);
            for (size_t i = j; i < n; i += len) {
                const size_t b = i + len / 2;
                const double tr = re[b] * c - im[b] * sn, ti = re[b] * sn + im[b] * c;
                re[b] = re[i] - tr;
                im[b] = im[i] - ti;
                re[i] += tr;
                im[i] += ti;
            }
        }
    }
}

void any_tables_free(AnyTables &t) {
    if (t.chirp) hipFree(t.chirp);
    if (t.midtab) hipFree(t.midtab);
    if (t.ws) hipFree(t.ws);
    if (t.mean) hipFree(t.mean);
    t = AnyTables{};
}

// tables of a length-nfft transform; the caller synchronises the stream before the host vectors die (done here)
int any_tables_init(oth_ctx *c, int nfft, AnyTables *t) {
    if (any_describe(nfft, &t->sh))
        return fail(c, OTH_ERR_UNSUPPORTED, "transform length outside [1, 1048576] (lengths that are not 2-3-5-7-smooth or exceed "
                                            "16384 without being a power of two run as Bluestein transforms of 2^ceil(log2(2 n - 1)) "
                                            "<= 1048576 points: n <= 524288)");
    int rc = get_twiddles(c, t->sh.L, &t->tw);
    if (rc) return rc;
    if (t->sh.kind == ANY_BLUESTEIN || t->sh.kind == ANY_BLUESTEIN2) {
        const int N = nfft, M = t->sh.L;
        std::vector<float2> ch(N), mt(M);
        std::vector<double> bre(M, 0.0), bim(M, 0.0);
        for (int n = 0; n < N; ++n) {
            const long long q = ((long long)n * (long long)n) % (2LL * N);      // the chirp's phase, reduced exactly
            const double a = M_PI * (double)q / (double)N;
            ch[n] = make_float2((float)cos(a), (float)-sin(a));                  // c[n] = exp(-i pi n^2 / N)
            bre[n] = cos(a);                                                     // b[n] = conj(c[n]), b[M - n] = b[n]
            bim[n] = sin(a);
            if (n) {
                bre[M - n] = bre[n];
                bim[M - n] = bim[n];
            }
        }
        host_fft_pow2(bre, bim);
        for (int k = 0; k < M; ++k) mt[k] = make_float2((float)(bre[k] / M), (float)(bim[k] / M));
        hipError_t e = hipMalloc(&t->chirp, sizeof(float2) * N);
        if (e == hipSuccess) e = hipMalloc(&t->midtab, sizeof(float2) * M);
        if (e == hipSuccess) e = hipMemcpyAsync(t->chirp, ch.data(), sizeof(float2) * N, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(t->midtab, mt.data(), sizeof(float2) * M, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            any_tables_free(*t);
            return fail(c, OTH_ERR_HIP, std::string("any-length tables: ") + hipGetErrorString(e));
        }
    }
    return OTH_OK;
}

// partial rows (per stream) an averaging launch of this shape leaves; pure host logic (the recipe text carries it)
int any_partial_rows(const AnyShape &sh, long long nseg, int cu_count) {
    long long w;
    if (sh.kind == ANY_DIRECT || sh.kind == ANY_BLUESTEIN) {
        // workgroups per CU by LDS footprint (tile + the staged twiddles where any_make_desc puts them there), at most 16:
        // the single-column builds hold ~100 registers, five 64-thread workgroups per SIMD
        const long long tile = (long long)sh.L * 8, both = 2 * tile;
        const long long lds = (both <= 64 * 1024 || (tile > 64 * 1024 && both <= 150 * 1024)) ? both : tile;
        long long occ = (150 * 1024) / lds;
        occ = occ < 1 ? 1 : (occ > 16 ? 16 : occ);
        w = (long long)cu_count * occ;
    } else {
        static const char *e = getenv("OTH_ANY_TL_W");      // (A/B of the partial-row count of the two-level routes)
        w = e && atoi(e) > 0 ? atoi(e) : 64;
    }
    if (w > nseg) w = nseg;
    if (w > 65535) w = 65535;
    return (int)(w < 1 ? 1 : w);
}

// workspace of a chunk: it and the chunk's samples stay in the Infinity Cache (OTH_ANY_WS_MB: A/B of the chunk size)
size_t any_ws_bytes() {
    const char *e = getenv("OTH_ANY_WS_MB");      // (read per call: the tests shrink it to cross chunk boundaries on small inputs)
    const long mb = e ? atol(e) : 0;
    return (size_t)(mb > 0 ? mb : 128) << 20;
}
constexpr int kAnyRowTile = 16;                // rows of L2 points a K2 workgroup transforms together

// nseg segments starting at x[first + s seg_step] (nperseg samples, window win, optional constant detrend) -> either the
// W x nch partial rows of |X|^2 (cross) sums (rows == nullptr; layout 0, or 6 = [k1][k2] for the two-level route), or one
// periodogram row per segment (rows != nullptr: epilogue / scale / fftshift as PgramArgs).  nbins = t.sh.nfft.
int any_run(oth_ctx *c, AnyTables &t, const float2 *x, const float2 *y, long long first, long long seg_step, int nperseg,
            const float *win, bool detrend, long long nseg, float *partial, int W, float *rows, int epilogue, float scale,
            int fftshift, bool coverage_only = false) {
    const AnyShape &sh = t.sh;
    const int nch = y ? 2 : 1, N = sh.nfft, L = sh.L;
    const bool two = sh.kind == ANY_TWOLEVEL || sh.kind == ANY_BLUESTEIN2, blu = sh.kind == ANY_BLUESTEIN || sh.kind == ANY_BLUESTEIN2;
    const int acc_store = y ? 2 : 1;
    long long B = two ? (long long)(any_ws_bytes() / (sizeof(float2) * (size_t)L * nch)) : (1LL << 20);
    if (B < 1) B = 1;
    if (B > nseg) B = nseg;
    int rc;
    if (two && (rc = ensure(c, &t.ws, &t.ws_cap, sizeof(float2) * (size_t)L * nch * (size_t)B))) return rc;
    // (the fast two-level route keeps sub-block sums there instead: at most B * seg_step / kTlSub + nperseg / kTlSub of them)
    const size_t nsums = nch * ((size_t)B * (size_t)(seg_step / kTlSub + 1) + (size_t)(nperseg / kTlSub) + 1);
    if (detrend && (rc = ensure(c, &t.mean, &t.mean_cap, sizeof(float4) * std::max(nch * (size_t)B, nsums)))) return rc;
    AnyFftDesc d_one{}, d_col{}, d_row{};
    if (two) {
        any_make_desc(sh.L1, sh.C, t.tw, L, &d_col);
        any_make_desc(sh.L2, kAnyRowTile, t.tw, L, &d_row);
    } else {
        any_make_desc(L, 1, t.tw, L, &d_one);
    }
    for (long long s0 = 0; s0 < nseg; s0 += B) {
        const long long nb = nseg - s0 < B ? nseg - s0 : B;
        const long long cfirst = first + s0 * seg_step;
        AnyArgs a{};
        // what every launch of the chunk shares
        a.nseg = nb;
        a.x = x;
        a.y = y;
        a.first = cfirst;
        a.seg_step = seg_step;
        a.nperseg = nperseg;
        a.win = win;
        a.mean = detrend ? t.mean : nullptr;
        a.mean_ch_stride = (size_t)B;
        a.ws = t.ws;
        a.ws_seg_stride = (size_t)L;
        a.ws_ch_stride = (size_t)L * (size_t)B;
        a.midtab = t.midtab;
        a.partial = partial;
        a.nbins = N;
        a.first_chunk = s0 == 0;
        a.conj_out = blu ? 1 : 0;
        a.rows = rows ? rows + (size_t)s0 * N : nullptr;
        a.epilogue = epilogue;
        a.scale = scale;
        a.fftshift = fftshift;
        const int gy_rows = (int)(nb < 65535 ? nb : 65535);
        if (sh.kind == ANY_TWOLEVEL && tl_supported(L) && !rows && !coverage_only) {
            // 32768 / 65536 points, averages (one or two channels): the register radix-16 kernels of fft_tl.hip
            const bool blocks = detrend && nperseg % kTlSub == 0 && seg_step % kTlSub == 0;      // (t.mean holds B float4 = B double2)
            TlArgs ta{};
            ta.x = x, ta.y = y, ta.first = cfirst, ta.seg_step = seg_step, ta.nperseg = nperseg, ta.win = win;
            ta.ws = t.ws, ta.ws_seg_stride = (size_t)L, ta.ws_ch_stride = (size_t)L * (size_t)B, ta.nseg = nb, ta.tw = t.tw;
            ta.partial = partial, ta.first_chunk = s0 == 0;
            if (blocks) {
                ta.nsub = nperseg / kTlSub, ta.sub_step = (int)(seg_step / kTlSub);
                const long long nblk = (nb - 1) * ta.sub_step + ta.nsub;
                ta.aux_ch_stride = (size_t)nblk;
                ta.bsum = reinterpret_cast<const double2 *>(t.mean);
                for (int ch = 0; ch < nch; ++ch)
                    HIPCHK(c, launch_tl_blocksum(ch ? y : x, cfirst, nblk, reinterpret_cast<double2 *>(t.mean) + (size_t)ch * nblk, c->stream));
            } else if (detrend) {
                ta.mean = t.mean;
                ta.aux_ch_stride = (size_t)B;
                for (int ch = 0; ch < nch; ++ch)
                    HIPCHK(c, launch_tl_mean(ch ? y : x, cfirst, seg_step, nperseg, nb, t.mean + (size_t)ch * B, c->stream));
            }
            HIPCHK(c, launch_tl_k1(L, ta, c->stream));
            HIPCHK(c, launch_tl_k2(L, ta, W, c->stream));
            continue;
        }
        if (detrend) HIPCHK(c, launch_any_mean(x, y, cfirst, seg_step, nperseg, nb, t.mean, (size_t)B, c->stream));
        if (!two) {
            // one launch: a workgroup per segment (rows W of the partial buffer), nothing leaves LDS
            a.f = d_one;
            a.es = 1, a.tile_stride = 0, a.cs = 1, a.inv_n = 1.0f / (float)L;
            a.load_op = 1;
            a.chirp = blu ? t.chirp : nullptr;
            a.mid_op = blu ? 1 : 0;
            a.nat_i = 1, a.pp_i = 1;
            HIPCHK(c, launch_any_fft(a, 1, rows ? gy_rows : W, 1, rows ? 3 : acc_store, c->stream));
            continue;
        }
        // K1: tiles of C columns (stride L2) of every segment, transform along L1, x W_L^(k1 n2), into the workspace
        AnyArgs k1 = a;
        k1.f = d_col;
        k1.es = sh.L2, k1.tile_stride = sh.C, k1.cs = 1, k1.inv_n = 1.0f / (float)sh.L1;
        k1.load_op = 1;
        k1.chirp = blu ? t.chirp : nullptr;
        k1.twbig = t.tw;
        k1.tw_t = sh.C, k1.tw_c = 1;
        HIPCHK(c, launch_any_fft(k1, sh.L2 / sh.C, gy_rows, nch, 0, c->stream));
        // K2: tiles of kAnyRowTile rows k1 (L2 contiguous points each), transform along L2: bins k1 + L1 k2
        AnyArgs k2 = a;
        k2.f = d_row;
        k2.es = 1, k2.tile_stride = kAnyRowTile * sh.L2, k2.cs = sh.L2, k2.inv_n = 1.0f / (float)sh.L2;
        k2.load_op = 0;
        k2.nat_i = sh.L1, k2.nat_t = kAnyRowTile, k2.nat_c = 1;
        if (!blu) {
            k2.pp_i = 1, k2.pp_t = kAnyRowTile * sh.L2, k2.pp_c = sh.L2;      // partial rows in [k1][k2] order (finalize layout 6)
            HIPCHK(c, launch_any_fft(k2, sh.L1 / kAnyRowTile, rows ? gy_rows : W, 1, rows ? 3 : acc_store, c->stream));
            continue;
        }
        // Bluestein: K2 = row transform, x B / M, conj, row transform, x W_L^(n2 k1), in place ...
        k2.mid_op = 1;
        k2.twbig = t.tw;
        k2.tw_t = kAnyRowTile, k2.tw_c = 1;
        HIPCHK(c, launch_any_fft(k2, sh.L1 / kAnyRowTile, gy_rows, nch, 0, c->stream));
        // ... K3 = column transform along L1 -> natural order n1 L2 + n2; the first nfft outputs are (conj of) X
        AnyArgs k3 = a;
        k3.f = d_col;
        k3.es = sh.L2, k3.tile_stride = sh.C, k3.cs = 1, k3.inv_n = 1.0f / (float)sh.L1;
        k3.load_op = 0;
        k3.nat_i = sh.L2, k3.nat_t = sh.C, k3.nat_c = 1;
        k3.pp_i = sh.L2, k3.pp_t = sh.C, k3.pp_c = 1;
        HIPCHK(c, launch_any_fft(k3, sh.L2 / sh.C, rows ? gy_rows : W, 1, rows ? 3 : acc_store, c->stream));
    }
    return OTH_OK;
}

// Forward transform of ONE length-L vector, natural order in and out, in place in `data` (L points); L is a power of two
// or smooth (kind ANY_DIRECT / ANY_TWOLEVEL of `sh`); tmp: L points (the two-level route's reordering).
int any_fft_nat_inner(oth_ctx *c, const AnyShape &sh, const float2 *tw, float2 *data, float2 *tmp) {
    AnyArgs a{};
    a.nseg = 1;
    a.ws = data;
    a.load_op = 0;
    if (sh.kind == ANY_DIRECT) {
        any_make_desc(sh.L, 1, tw, sh.L, &a.f);
        a.es = 1, a.cs = 1, a.inv_n = 1.0f / (float)sh.L;
        HIPCHK(c, launch_any_fft(a, 1, 1, 1, 0, c->stream));
        return OTH_OK;
    }
    AnyArgs k1 = a;
    any_make_desc(sh.L1, sh.C, tw, sh.L, &k1.f);
    k1.es = sh.L2, k1.tile_stride = sh.C, k1.cs = 1, k1.inv_n = 1.0f / (float)sh.L1;
    k1.twbig = tw;
    k1.tw_t = sh.C, k1.tw_c = 1;
    HIPCHK(c, launch_any_fft(k1, sh.L2 / sh.C, 1, 1, 0, c->stream));
    AnyArgs k2 = a;
    any_make_desc(sh.L2, kAnyRowTile, tw, sh.L, &k2.f);
    k2.es = 1, k2.tile_stride = kAnyRowTile * sh.L2, k2.cs = sh.L2, k2.inv_n = 1.0f / (float)sh.L2;
    HIPCHK(c, launch_any_fft(k2, sh.L1 / kAnyRowTile, 1, 1, 0, c->stream));
    HIPCHK(c, launch_any_ew(3, tmp, data, nullptr, nullptr, sh.L, sh.L, sh.L1, sh.L2, c->stream));      // [k1][k2] -> k1 + L1 k2
    HIPCHK(c, hipMemcpyAsync(data, tmp, sizeof(float2) * (size_t)sh.L, hipMemcpyDeviceToDevice, c->stream));
    return OTH_OK;
}

// points of scratch any_fft_nat() needs behind the data
size_t any_fft_nat_scratch(const AnyShape &sh) { return 2 * (size_t)sh.L; }

// np.fft.fft of one length-nfft vector in `data` (natural order, in place) for every route; scratch as above
int any_fft_nat(oth_ctx *c, const AnyTables &t, float2 *data, float2 *scratch) {
    const AnyShape &sh = t.sh;
    if (sh.kind == ANY_DIRECT || sh.kind == ANY_TWOLEVEL) return any_fft_nat_inner(c, sh, t.tw, data, scratch);
    AnyShape in{};
    if (any_describe(sh.L, &in)) return fail(c, OTH_ERR_INTERNAL, "Bluestein length has no route");
    float2 *A = scratch, *tmp = scratch + sh.L;
    int rc;
    HIPCHK(c, launch_any_ew(0, A, data, nullptr, t.chirp, sh.L, sh.nfft, 0, 0, c->stream));      // a = x c, zero padded to M
    if ((rc = any_fft_nat_inner(c, in, t.tw, A, tmp))) return rc;
    HIPCHK(c, launch_any_ew(1, A, A, nullptr, t.midtab, sh.L, sh.L, 0, 0, c->stream));            // conj(A B / M)
    if ((rc = any_fft_nat_inner(c, in, t.tw, A, tmp))) return rc;
    HIPCHK(c, launch_any_ew(2, data, A, nullptr, t.chirp, sh.nfft, sh.nfft, 0, 0, c->stream));    // X = conj(.) c
    return OTH_OK;
}

// ---- routing as data (round 5) ---------------------------------------------------------------------------------------
// Which kernel build runs a launch, with which detrend form, pilot, schedule, chunk sizes, grid and partial-row layout,
// is decided by resolve_recipe() from the plan's shape and the launch's segment count - pure host logic, no HIP call, so
// tests/test_abi_cpu.py::test_launch_recipes_table can enumerate it without a GPU through oth__debug_recipe().
// run_average() below only allocates, fills the argument structs and dispatches on the recipe.
enum RecipeKernel {
    RK_GENERIC = 0,      // welch_generic_kernel (coverage Stockham kernel; also the two-channel coverage path)
    RK_W4096,            // welch4096[ws]_kernel: the build in `variant`
    RK_CSD4096,          // csd4096_kernel (one role)
    RK_CSD4096WS,        // csd4096ws_kernel (role-split pairs)
    RK_W16K,             // welch16k_kernel<., F> (4 x 4096 / 2 x 4096)
    RK_W16K1X,           // welch16k1x_pipe_kernel / welch16k1x_kernel (one exchange, no overlap)
    RK_W16K1X_HALF,      // welch16k1x_half_kernel (one exchange, 50 % overlap)
    RK_SEG,              // seg_kernel<R, ...>
    RK_SEGWS,            // segws_kernel<R, DET>
    RK_SEGPAD,           // seg_kernel<R, ..., NA> zero-padded
    RK_ANY,              // any_fft_kernel launches (fft_any.hip): lengths the kernels above do not take
};
const char *const kRecipeKernelName[] = {"welch_generic", "welch4096", "csd4096", "csd4096ws", "welch16k", "welch16k1x",
                                         "welch16k1x_half", "seg", "segws", "seg_padded", "anyfft"};
const char *const kAnyKindName[] = {"none", "direct", "twolevel", "bluestein", "bluestein2"};

// what resolve_recipe() needs to know of a plan (oth_plan holds the same fields; the debug entry builds one by hand)
struct PlanShape {
    int nfft = 0, nperseg = 0, step = 0;
    bool detrend = false, fast_detrend = false;
    bool fd_ok = false;          // a window-spectrum table exists for this size (oth_plan::d_fd)
    bool fd1x_ok = false;        // ... and the one for welch16k1x_half (spectrum confined to |k| < 16)
    bool rect_window = false;
    int kernel = OTH_KERNEL_AUTO, sched = OTH_SCHED_DYNAMIC;
    bool pilot_launch = false;
    std::string tune_variant;
    int tune_sched = -1, tune_chunk = 0, tune_tail = 0;
    AnyShape any{};              // kind != ANY_NONE: the any-length route
};

// resident workgroups (teams) per CU of a tuned build: the runtime asks the occupancy calculator (needs a device), the
// debug entry uses the table below - what the calculator returns on MI355X for the shipped builds
// (tests/test_hip_parity.py::test_recipe_occupancy_table_matches_the_runtime compares the two on the GPU)
struct OccupancyKey {
    RecipeKernel kern;
    const char *variant;      // RK_W4096
    int nfft, nperseg, seg_kind;
    bool seg_wps4;
    bool half_ws;             // RK_W16K1X_HALF at 8192 points: the role-split build (one 1024-thread workgroup per CU)
};
int runtime_bpc(const OccupancyKey &k) {
    switch (k.kern) {
        case RK_W4096: return variant_by_tag(k.variant)->blocks_per_cu();
        case RK_CSD4096: return csd4096_blocks_per_cu();
        case RK_CSD4096WS: return csd4096ws_blocks_per_cu();
        case RK_SEG: return seg_teams_per_cu(k.nfft, k.seg_kind, k.seg_wps4);
        case RK_SEGWS: return segws_teams_per_cu(k.nfft);
        case RK_SEGPAD: return seg_padded_teams_per_cu(k.nfft, k.nperseg, k.seg_kind);
        case RK_W16K: case RK_W16K1X: case RK_W16K1X_HALF: return k.nfft == 8192 && !k.half_ws ? 2 : 1;      // 70 / 139 (145) KiB of LDS
        default: return 0;
    }
}
int table_bpc(const OccupancyKey &k) {
    switch (k.kern) {
        case RK_W4096: return !strcmp(k.variant, "ws") ? 2 : 4;      // 78 KiB of LDS / 35 KiB and 128 VGPRs
        case RK_CSD4096: return 3;
        case RK_CSD4096WS: return 1;
        case RK_SEGWS: return k.nfft == 1024 ? 8 : 4;
        case RK_SEG: {      // teams per CU: waves per SIMD (3; "seg4": 4) x 4 SIMDs x teams per wave / waves per team
            const int tpw = k.nfft == 256 ? 4 : (k.nfft == 512 ? 2 : 1), wpt = k.nfft <= 1024 ? 1 : k.nfft / 1024;
            return 4 * (k.seg_wps4 ? 4 : 3) * tpw / wpt;
        }
        case RK_SEGPAD:      // 2048: NA = 4 at four waves per SIMD; NA = 8 at three, where the half-load pilot build takes 130 VGPRs
            return k.nfft == 1024 ? 16 : (k.nperseg * 4 == k.nfft ? 8 : (k.seg_kind == 0 ? 6 : 8));
        case RK_W16K: case RK_W16K1X: case RK_W16K1X_HALF: return k.nfft == 8192 && !k.half_ws ? 2 : 1;
        default: return 0;
    }
}

struct LaunchRecipe {
    RecipeKernel kern = RK_GENERIC;
    const W4096Variant *variant = nullptr;      // RK_W4096
    bool csd = false;
    int form = 0;                // constant detrend: 0 none, 1 before the window (time domain), 2 after the transform (needs the table)
    int pilot = 0;               // 0 none (no detrend, or OTH_DETREND_CONSTANT_FAST), 1 pilot_mean_kernel in front, 2 in the kernel's prologue
    bool use_fd1x = false;       // the table handed to the kernel is d_fd1x
    int seg_kind = 0, seg_det = 0;
    bool seg_wps4 = false;
    bool x1_window = false, x1_plain = false;      // RK_W16K1X: windowed build / the un-pipelined loop
    bool half_ws = false;                          // RK_W16K1X_HALF, 8192 points: welch8kws_kernel
    int bpc = 0;                 // resident workgroups (teams) per CU (0: generic grid rule)
    int W = 1, rows = 1, nch = 1, layout = 0;
    int sched = 0, chunk = 1, tail_chunk = 1;
    long long nbig = 0, nseg_run = 0;
    bool tickets = false;        // draws chunk tickets from the context's queue
    bool two_runs = false;       // "ws2": the stream cut into two runs of segments
    int any_kind = 0;            // RK_ANY: AnyKind
    bool any_r16 = false;        // ... on fft_tl.hip's register radix-16 kernels (32768 / 65536 points)
    bool any_onewg = false;      // ... 32768 points, one channel, full segments: welch32k.hip (the segment never leaves the CU)
};

int generic_wg_for(int cu_count, int nfft, long long nseg, int nstreams) {
    const size_t lds = generic_lds_bytes(nfft);
    long long occ = (long long)(160 * 1024 / lds);
    const long long tocc = 2048 / generic_threads_for(nfft);
    if (occ > tocc) occ = tocc;
    if (occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    long long w = ((long long)cu_count * occ + nstreams - 1) / nstreams;
    if (w > nseg) w = nseg;
    if (w < 1) w = 1;
    return (int)w;
}

// -> OTH_OK, or OTH_ERR_UNSUPPORTED with *why set (OTH_KERNEL_TUNED on a plan no tuned kernel covers)
int resolve_recipe(const PlanShape &p, bool csd, long long nseg, int nstreams, int cu_count, int (*bpc_of)(const OccupancyKey &),
                   LaunchRecipe *out, const char **why) {
    LaunchRecipe r;
    r.csd = csd;
    r.nch = csd ? 4 : 1;
    if (p.any.kind != ANY_NONE) {
        // lengths outside the power-of-two kernels: detrend in the time domain from each segment's own mean (no pilot),
        // contiguous strided rows of segments, partial rows in natural order (two-level: [k1][k2], finalize layout 6)
        if (p.kernel == OTH_KERNEL_TUNED) {
            if (why) *why = "tuned kernel does not cover this plan";
            return OTH_ERR_UNSUPPORTED;
        }
        r.kern = RK_ANY;
        r.any_kind = p.any.kind;
        r.any_onewg = p.any.kind == ANY_TWOLEVEL && (p.any.L == 32768 || p.any.L == 65536) && p.nperseg == p.any.L && !csd &&
                      p.tune_variant != "anycov" && p.tune_variant != "r16";                             // welch32k.hip
        r.any_r16 = !r.any_onewg && p.any.kind == ANY_TWOLEVEL && tl_supported(p.any.L) && p.tune_variant != "anycov";      // fft_tl.hip
        r.form = p.detrend ? 1 : 0;
        r.W = r.any_onewg ? welch32k_rows(nseg, cu_count, p.any.L == 65536) : any_partial_rows(p.any, nseg, cu_count);
        r.layout = r.any_onewg ? (p.any.L == 65536 ? 8 : 7) : (p.any.kind == ANY_TWOLEVEL ? 6 : 0);
        r.nseg_run = nseg;
        *out = r;
        return OTH_OK;
    }
    const std::string &tv = p.tune_variant;
    const bool want_tuned = p.kernel != OTH_KERNEL_GENERIC;
    const bool half_step = p.step * 2 == p.nperseg;
    // ---- detrend form.  After the transform (FFT((x - m) w) = FFT(x w) - m FFT(w), the role-split / half-keeping
    // builds) only with a window-spectrum table and at least kFdMinSegments segments per stream: below that the
    // time-domain builds run in both detrend modes (the pilot is one value per launch; an offset that moves within a
    // one- or two-segment launch has nothing to average its rounding down - advisor, round 4 - and such launches do not
    // need the fast builds' throughput).  A variant forced through oth_plan_set_tuning is honoured; "td" forces the
    // time-domain builds at any length.
    const bool fd_forced = !tv.empty() && tv != "td" && tv != "plaunch";
    const bool fd = p.fd_ok && tv != "td" && (fd_forced || nseg >= kFdMinSegments);
    const bool fd1x = fd && p.fd1x_ok;
    // ---- kernel family
    const bool pow2_nperseg = p.nperseg >= 256 && (p.nperseg & (p.nperseg - 1)) == 0;
    const bool seg_size = p.nfft == 256 || p.nfft == 512 || p.nfft == 1024 || p.nfft == 2048;
    const bool big_size = p.nfft == 8192 || p.nfft == 16384;
    if (csd) {
        if (want_tuned && p.nfft == 4096 && p.nperseg == 4096) {
            // role-split pairs: 50 % overlap, frequency-domain detrend, 32-bit segment indices; "csd1" forces the one-role kernel
            const bool ws = p.step == 2048 && (!p.detrend || fd) && nseg < (1LL << 30) && tv != "csd1";
            r.kern = ws ? RK_CSD4096WS : RK_CSD4096;
        }
    } else if (want_tuned && p.nfft == 4096 && pow2_nperseg) {      // nperseg = 256 ... 4096, zero-padded to 4096
        r.kern = RK_W4096;
        const bool fd_ok = (!p.detrend || fd) && nseg < (1LL << 30);      // ws: 32-bit segment indices
        r.variant = w4096_variant(p.nperseg == 4096 ? p.step : 0, fd_ok, tv);
        // "ws2" cuts the stream into two equal runs of segments: an odd count (or a single segment) stays on "ws"
        if (!strcmp(r.variant->tag, "ws2") && (nseg < 2 || (nseg & 1))) r.variant = variant_by_tag("ws");
        r.two_runs = !strcmp(r.variant->tag, "ws2");
    } else if (want_tuned && big_size && (p.nperseg == p.nfft || p.nperseg * 4 == p.nfft)) {      // (nfft / 4: the sweeper's zero padding)
        r.kern = RK_W16K;
        if (p.nperseg == p.nfft && tv != "16k4") {
            // one cross-wave exchange per segment (16384: 16 waves, one workgroup per CU; 8192, round 5: 8 waves, two
            // radix-8 butterflies in pass 2, two workgroups per CU): vectors that do not overlap without a detrend (the
            // scanner of BASELINE config 5; at 8192 points the pipelined rectangular build only), and 50 % overlap with
            // the kept half in registers (a constant detrend needs the |k| < 16 table)
            if (p.step >= p.nfft && !p.detrend && (p.nfft == 16384 || (p.rect_window && tv != "16kplain"))) r.kern = RK_W16K1X;
            else if (p.step * 2 == p.nfft && (!p.detrend || fd1x)) {
                r.kern = RK_W16K1X_HALF;
                // the role-split build at 8192 points walks contiguous runs only (what this shape takes by default)
                const bool contiguous = (p.tune_sched < 0 && p.sched == OTH_SCHED_DYNAMIC) ||
                                        (p.tune_sched >= 0 ? p.tune_sched : p.sched) == OTH_SCHED_CONTIGUOUS;
                r.half_ws = p.nfft == 8192 && contiguous && nseg < (1LL << 31) && tv != "8k1role";      // ("8k1role": the A/B)
            }
        }
    } else if (want_tuned && seg_size && seg_padded_supported(p.nfft, p.nperseg)) {
        r.kern = RK_SEGPAD;      // nperseg = nfft / 4 (the sweeper's call, spectrum_sweeper.py:263) or nfft / 2 at 1024 / 2048
    } else if (want_tuned && seg_size && p.nperseg == p.nfft) {
        r.kern = RK_SEG;
    }
    if (p.kernel == OTH_KERNEL_TUNED && r.kern == RK_GENERIC) {
        if (why) *why = "tuned kernel does not cover this plan";
        return OTH_ERR_UNSUPPORTED;
    }
    if (r.kern == RK_SEG || r.kern == RK_SEGPAD) {
        r.seg_kind = half_step ? 0 : 1;
        r.seg_wps4 = tv == "seg4";
        // role-split build: 50 % overlap; detrend in the time domain at 1024 (one producer wave), after the transform at
        // 2048 (needs the table); "seg3" / "seg4" force the one-role builds
        r.seg_det = !p.detrend ? 0 : (p.nfft == 1024 ? 1 : 2);
        if (r.kern == RK_SEG && p.nfft >= 1024 && r.seg_kind == 0 && tv != "seg3" && !r.seg_wps4 && (r.seg_det != 2 || fd))
            r.kern = RK_SEGWS;
    }
    // ---- detrend form and pilot of the kernel chosen
    if (p.detrend) {
        const bool after = r.kern == RK_CSD4096WS || r.kern == RK_W16K1X_HALF || (r.kern == RK_W4096 && r.variant->fd) ||
                           (r.kern == RK_SEGWS && r.seg_det == 2) || (r.kern == RK_W16K && fd && half_step && p.nperseg == p.nfft);
        r.form = after ? 2 : 1;
        r.use_fd1x = r.kern == RK_W16K1X_HALF;
        if (!p.fast_detrend) {
            const bool can_inline = ((r.kern == RK_W4096 && r.variant->inline_pilot) || r.kern == RK_CSD4096WS ||
                                     r.kern == RK_W16K1X_HALF) &&
                                    !p.pilot_launch && tv != "plaunch";
            r.pilot = can_inline ? 2 : 1;
        }
    }
    if (r.kern == RK_W16K1X) {
        r.x1_window = !p.rect_window;
        r.x1_plain = tv == "16kplain";
    }
    // ---- grid: exactly the resident workgroups (one wave of workgroups, no tail round); generic: by LDS footprint
    r.rows = r.kern == RK_W4096 ? r.variant->rows : 1;
    r.nseg_run = r.two_runs ? nseg / 2 : nseg;      // segments the schedule of one run covers
    if (r.kern == RK_GENERIC) {
        r.W = generic_wg_for(cu_count, p.nfft, nseg, nstreams);
    } else {
        const OccupancyKey key{r.kern, r.kern == RK_W4096 ? r.variant->tag : "", p.nfft, p.nperseg, r.seg_kind, r.seg_wps4, r.half_ws};
        r.bpc = bpc_of(key);
        if (r.bpc < 1) r.bpc = 1;
        const long long w = ((long long)cu_count * r.bpc + nstreams - 1) / nstreams;
        r.W = (int)(w > r.nseg_run ? r.nseg_run : (w < 1 ? 1 : w));
    }
    r.layout = (r.kern == RK_W4096 || r.kern == RK_CSD4096 || r.kern == RK_CSD4096WS) ? 1
               : (r.kern == RK_W16K1X || r.kern == RK_W16K1X_HALF) ? (p.nfft == 16384 ? 4 : 5)
               : (r.kern == RK_W16K ? (p.nfft == 16384 ? 2 : 3) : 0);
    // ---- schedule and chunks (tuned kernels only; the coverage kernel walks contiguous runs)
    if (r.kern != RK_GENERIC) {
        const bool auto_sched = p.tune_sched < 0 && p.sched == OTH_SCHED_DYNAMIC;      // "the library's choice"
        const long long per_team = r.nseg_run / (r.W > 0 ? r.W : 1);
        const bool is_seg = r.kern == RK_SEG || r.kern == RK_SEGPAD || r.kern == RK_SEGWS;
        const bool big = r.kern == RK_W16K || r.kern == RK_W16K1X || r.kern == RK_W16K1X_HALF;
        r.sched = p.tune_sched >= 0 ? p.tune_sched : p.sched;
        int static_chunk = 0;
        if (auto_sched) {
            // one 1024-thread workgroup per CU and equal work per segment: contiguous runs beat the ticket queue (+3 %)
            if (r.kern == RK_CSD4096WS) r.sched = OTH_SCHED_CONTIGUOUS;
            // the role-split 1024 kernel: eight two-wave workgroups per CU even out by themselves (+4 % over the tickets)
            if (r.kern == RK_SEGWS && p.nfft == 1024 && nstreams == 1) r.sched = OTH_SCHED_CONTIGUOUS;
            if (r.kern == RK_SEG || r.kern == RK_SEGPAD) {
                // 256 / 512 points (and the zero-padded builds: nfft / 8 new samples per segment) at 50 % overlap: a ticket
                // per sixteen 2-4 KiB segments costs more than it evens out (17-35 % of the roofline at every launch size).
                // Static instead: interleaved chunks of 32 / 16 segments while every team gets two of them (256 points,
                // 2^27 samples: 66 % against 43 %), one contiguous run per team below that.
                if (r.seg_kind == 0 && (p.nfft <= 512 || r.kern == RK_SEGPAD)) {
                    static_chunk = per_team >= 64 ? 32 : (per_team >= 32 ? 16 : 0);
                    r.sched = static_chunk ? OTH_SCHED_INTERLEAVED : OTH_SCHED_CONTIGUOUS;
                }
                // whole-segment loads (steps other than nfft / 2): the next chunk's first segment is prefetched across the
                // chunk boundary only under the interleaved schedule - 8-segment chunks: 1024 points, no overlap, 70 % of
                // the roofline against 49 % with tickets
                if (r.seg_kind == 1) {
                    static_chunk = per_team >= 16 ? 8 : 0;
                    r.sched = static_chunk ? OTH_SCHED_INTERLEAVED : OTH_SCHED_CONTIGUOUS;
                }
            }
            // short launches (fewer than 32 segments per resident workgroup): one contiguous run each - the tickets' guided
            // tail has nothing to even out and costs 5-20 % (2048 points, 2^22 samples: 17.3 % against 14.0 %)
            if ((r.kern == RK_W4096 || r.kern == RK_SEGWS) && per_team < 32) r.sched = OTH_SCHED_CONTIGUOUS;
            // the one-exchange 16384-point scanner kernel: one workgroup per CU, equal work per segment, no chunk head to
            // re-read - contiguous runs (0.417-0.418 against 0.421-0.424 ms with tickets, same box) unless a workgroup gets
            // so few segments that an uneven split shows
            if (r.kern == RK_W16K1X && per_team >= 8) r.sched = OTH_SCHED_CONTIGUOUS;
            // 16384 points at 50 % overlap: contiguous runs (no chunk head is read twice): 30.8 % against 29.3 % with tickets
            if (big && p.nfft == 16384 && half_step && p.nperseg == p.nfft) r.sched = OTH_SCHED_CONTIGUOUS;
            // the one-exchange 50 %-overlap build prefetches across its run, and a chunk head costs it a synchronous load
            if (r.kern == RK_W16K1X_HALF) r.sched = OTH_SCHED_CONTIGUOUS;
            // 8192 (round 4): contiguous runs take the same time as tickets over chunks of 16 and read no chunk head twice
            if (big && p.nfft == 8192 && half_step && p.nperseg == p.nfft && per_team >= 16) r.sched = OTH_SCHED_CONTIGUOUS;
        }
        if (r.sched < 0 || r.sched > 2) r.sched = 0;
        // segments per chunk
        int chunk;
        if (p.tune_chunk > 0) chunk = p.tune_chunk;
        else if (big) {
            const bool halves = half_step && (p.nperseg == p.nfft || p.nperseg * 4 == p.nfft);      // a kept half: longer chunks
            chunk = halves ? 16 : 2;
        } else if (r.kern == RK_W4096) chunk = r.variant->chunk;
        else if (is_seg) chunk = static_chunk ? static_chunk
                                              : (((p.nfft == 1024 && r.kern != RK_SEGWS) || (p.nfft == 2048 && r.kern == RK_SEGWS)) ? 32 : 16);
        else chunk = 8;      // the one-role two-channel kernel
        if (chunk < 1) chunk = 1;
        // welch16k1x: the ticket for the NEXT chunk is published with a chunk's first segment and read at its last
        if (r.kern == RK_W16K1X && chunk < 2) chunk = 2;
        r.chunk = chunk;
        r.tail_chunk = chunk;
        r.nbig = r.nseg_run / chunk;
        if (r.sched == OTH_SCHED_DYNAMIC) {
            if (nstreams > 64) {
                r.sched = OTH_SCHED_INTERLEAVED;      // the context holds 64 ticket words
            } else {
                r.tickets = true;
                // guided tail: the last half round of work goes out in quarter-size chunks
                r.tail_chunk = p.tune_tail > 0 ? p.tune_tail : (chunk >= 4 ? chunk / 4 : 1);
                if (r.tail_chunk < 1) r.tail_chunk = 1;
                if (r.kern == RK_W16K1X && r.tail_chunk < 2) r.tail_chunk = 2;
                const long long tail_segs = (long long)r.W * chunk / 2;
                r.nbig = r.nseg_run > tail_segs ? (r.nseg_run - tail_segs) / chunk : 0;
            }
        }
    }
    *out = r;
    return OTH_OK;
}

// the recipe as text (oth__debug_recipe, bench.py's kernel labels)
std::string recipe_text(const LaunchRecipe &r, int nfft) {
    static const char *const kForm[] = {"none", "time", "freq"}, *const kPilot[] = {"none", "launch", "inline"},
                             *const kSched[] = {"contiguous", "interleaved", "dynamic"};
    char buf[384];
    std::string k = kRecipeKernelName[r.kern];
    if (r.kern == RK_W4096) k += std::string(":") + r.variant->tag;
    if (r.kern == RK_SEG) k += std::string(r.seg_kind ? ":full" : ":half") + (r.seg_wps4 ? ":wps4" : "");
    if (r.kern == RK_SEGPAD) k += r.seg_kind ? ":full" : ":half";
    if (r.kern == RK_W16K1X) k += std::string(r.x1_plain || r.x1_window ? ":plain" : ":pipe") + (r.x1_window ? ":window" : "");
    if (r.kern == RK_W16K1X_HALF && r.half_ws) k += ":ws";
    if (r.kern == RK_ANY) k += r.any_onewg ? std::string(":onewg") : std::string(":") + kAnyKindName[r.any_kind] + (r.any_r16 ? ":r16" : "");
    snprintf(buf, sizeof buf, "kernel=%s nfft=%d form=%s pilot=%s sched=%s chunk=%d tail=%d nbig=%lld bpc=%d W=%d rows=%d nch=%d layout=%d",
             k.c_str(), nfft, kForm[r.form], kPilot[r.pilot], kSched[r.sched], r.chunk, r.tail_chunk, r.nbig, r.bpc, r.W, r.rows,
             r.nch, r.layout);
    return buf;
}

PlanShape shape_of(const oth_plan *p) {
    PlanShape s;
    s.nfft = p->nfft;
    s.nperseg = p->nperseg;
    s.step = p->step;
    s.detrend = p->detrend != OTH_DETREND_NONE;
    s.fast_detrend = p->fast_detrend;
    s.fd_ok = p->d_fd != nullptr;
    s.fd1x_ok = p->d_fd1x != nullptr;
    s.rect_window = p->rect_window;
    s.kernel = p->kernel;
    s.sched = p->sched;
    s.pilot_launch = p->pilot_launch;
    s.tune_variant = p->tune_variant;
    s.tune_sched = p->tune_sched;
    s.tune_chunk = p->tune_chunk;
    s.tune_tail = p->tune_tail;
    s.any = p->any.sh;
    return s;
}

// Launch the averaging kernel: partial sums land in plan->d_partial.
int run_average(oth_plan *p, const float2 *x, const float2 *y, size_t nsamples, int nstreams, size_t stride,
                long long *nseg_out, int *W_out, int *layout_out) {
    oth_ctx *c = p->ctx;
    long long nseg = 0;
    if (segments(p, nsamples, &nseg) != OTH_OK)
        return fail(c, OTH_ERR_INVALID, "input shorter than nperseg");
    const bool csd = (y != nullptr);
    LaunchRecipe r;
    const char *why = "";
    if (int rrc = resolve_recipe(shape_of(p), csd, nseg, nstreams, c->cu_count, runtime_bpc, &r, &why))
        return fail(c, rrc, why);
    p->last_recipe = recipe_text(r, p->nfft);
    // + 1 KiB per row of stamp space behind the sums (only the diagnostic kernel builds write it)
    int rc = ensure(c, &p->d_partial, &p->partial_cap,
                    sizeof(float) * (size_t)nstreams * r.W * r.rows * r.nch * p->nfft + 2048 * (size_t)nstreams * r.W * r.rows);      // (+ 2 KiB per row: the phase stamps of the diagnostic builds)
    p->last_W = r.W * r.rows * nstreams;
    {
        const int groups = std::max(kReduceGroups, finalize_row_groups(p->nfft, r.W * r.rows, r.nch));
        if (!rc) rc = ensure(c, &p->d_reduce, &p->reduce_cap, sizeof(float) * (size_t)nstreams * groups * r.nch * p->nfft);
    }
    if (rc) return rc;
    const float4 *fd_tab = r.form == 2 ? (r.use_fd1x ? p->d_fd1x : p->d_fd) : nullptr;
    // the pilot of every stream (WelchArgs.pilot): from its own launch, or formed in the kernel's prologue
    const float2 *pilot = nullptr;
    if (r.pilot == 1) {
        rc = ensure(c, &p->d_pilot, &p->pilot_cap, sizeof(float2) * 2 * kPilotProbes * (size_t)nstreams);
        if (rc) return rc;
        HIPCHK(c, launch_pilot_mean(x, csd ? y : nullptr, stride, p->nperseg, p->step, nseg, nstreams, p->d_pilot, c->stream));
        pilot = p->d_pilot;
    }
    unsigned *queue = nullptr;
    if (r.tickets) {
        queue = c->queue;
        if (!c->queue_clean) HIPCHK(c, hipMemsetAsync(c->queue, 0, sizeof(unsigned) * 64, c->stream));
        c->queue_clean = false;      // until the finalize launch that follows has re-zeroed them
        c->queue_used = nstreams;
    }
    if (r.kern == RK_ANY && r.any_onewg) {
        Timed tm(c);
        for (int st = 0; st < nstreams; ++st) {
            W32kArgs a{};
            a.x = x + (size_t)st * stride;
            a.first = 0, a.step = p->step, a.nseg = nseg;
            a.win = p->d_win, a.tw = p->any.tw;
            a.partial = p->d_partial + (size_t)st * r.W * p->nfft;
            a.detrend = p->detrend != OTH_DETREND_NONE;
            a.front = p->nfft == 65536, a.wpm = p->d_wpm;
            HIPCHK(c, launch_welch32k(a, r.W, c->stream));
        }
    } else if (r.kern == RK_ANY) {
        Timed tm(c);
        for (int st = 0; st < nstreams; ++st) {
            rc = any_run(c, p->any, x + (size_t)st * stride, csd ? y + (size_t)st * stride : nullptr, 0, p->step, p->nperseg, p->d_win,
                         p->detrend != OTH_DETREND_NONE, nseg, p->d_partial + (size_t)st * r.W * r.nch * p->nfft, r.W, nullptr, 0,
                         1.0f, 0, p->tune_variant == "anycov");
            if (rc) return rc;
        }
    } else if (r.kern == RK_SEG || r.kern == RK_SEGWS || r.kern == RK_SEGPAD) {
        SegArgs g{};
        g.x = x;
        g.stream_stride = stride;
        g.nstreams = nstreams;
        g.win = p->d_win;
        g.tw = p->d_tw;
        g.first = 0;
        g.step = p->step;
        g.nseg = nseg;
        g.detrend = p->detrend;
        g.chain = 0;
        g.partial = p->d_partial;
        g.wg_per_stream = r.W;
        g.sched = r.sched;
        g.chunk = r.chunk;
        g.tail_chunk = r.tail_chunk;
        g.nbig = r.nbig;
        g.queue = queue;
        g.fd = fd_tab;
        g.pilot = pilot;
        Timed tm(c);
        switch (r.kern) {
            case RK_SEGPAD: HIPCHK(c, launch_seg_padded(p->nfft, p->nperseg, g, r.seg_kind, c->stream)); break;
            case RK_SEGWS: HIPCHK(c, launch_segws(p->nfft, g, r.seg_det, c->stream)); break;
            default: HIPCHK(c, launch_seg(p->nfft, g, r.seg_kind, r.seg_wps4, c->stream)); break;
        }
    } else {
        WelchArgs a;
        a.x = x;
        a.y = y;
        a.win = p->d_win;
        a.tw = p->d_tw;
        a.partial = p->d_partial;
        a.nseg = r.nseg_run;
        if (r.two_runs) a.y = x + (size_t)r.nseg_run * 2048;      // run B starts nseg / 2 segments in (its first half-block is run A's last)
        a.stream_stride = stride;
        a.nperseg = p->nperseg;
        a.step = p->step;
        a.detrend = p->detrend;
        a.wg_per_stream = r.W;
        a.nstreams = nstreams;
        a.sched = r.sched;
        a.chunk = r.chunk;
        a.tail_chunk = r.tail_chunk;
        a.nbig = r.nbig;
        a.queue = queue;
        a.fd = fd_tab;
        a.pilot = pilot;
        a.pilot_inline = r.pilot == 2 ? 1 : 0;
        Timed tm(c);
        switch (r.kern) {
            case RK_W4096: HIPCHK(c, r.variant->launch(a, c->stream)); break;
            case RK_CSD4096WS: HIPCHK(c, launch_csd_tuned4096ws(a, c->stream)); break;
            case RK_CSD4096: HIPCHK(c, launch_csd_tuned4096(a, c->stream)); break;
            case RK_W16K1X_HALF:
                HIPCHK(c, r.half_ws ? launch_welch_tuned8kws(a, c->stream) : launch_welch_tuned16k1x_half(p->nfft, a, c->stream));
                break;
            case RK_W16K1X: HIPCHK(c, launch_welch_tuned16k1x(p->nfft, a, r.x1_window, r.x1_plain, c->stream)); break;
            case RK_W16K: HIPCHK(c, launch_welch_tuned16k(p->nfft, a, c->stream)); break;
            default: HIPCHK(c, launch_welch_generic(p->nfft, a, c->stream)); break;
        }
    }
    *nseg_out = nseg;
    *W_out = r.W * r.rows;
    *layout_out = r.layout;
    return OTH_OK;
}

// finalize_kernel zeroes the ticket counters the averaging launch before it used (saves a memset per call)
int finalize_and_rearm(oth_ctx *c, FinalizeArgs &f, int nstreams) {
    f.queue_reset = nullptr;
    f.queue_n = 0;
    if (!c->queue_clean && c->queue_used > 0) {
        f.queue_reset = c->queue;
        f.queue_n = c->queue_used;
    }
    HIPCHK(c, launch_finalize(f, nstreams, c->stream));
    if (f.queue_reset) {
        c->queue_clean = true;
        c->queue_used = 0;
    }
    return OTH_OK;
}

}  // namespace

extern "C" {

int oth_abi_version(void) { return OTH_ABI_VERSION; }

const char *oth_strerror(int code) {
    switch (code) {
        case OTH_OK: return "ok";
        case OTH_ERR_INVALID: return "invalid argument";
        case OTH_ERR_HIP: return "HIP runtime error / no usable GPU";
        case OTH_ERR_UNSUPPORTED: return "unsupported size or mode";
        case OTH_ERR_NOMEM: return "out of memory (device or host)";
        case OTH_ERR_STATE: return "invalid call order";
        case OTH_ERR_INTERNAL: return "internal error (C++ exception caught at the ABI)";
        default: return "unknown error";
    }
}

int oth_device_count(int *count) {
    OTH_TRY
    if (!count) return fail(nullptr, OTH_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, OTH_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return OTH_OK;
    OTH_CATCH(nullptr)
}

static int ctx_create(int device_id, void *stream, bool adopt, oth_ctx **out) {
    if (!out) return fail(nullptr, OTH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, OTH_ERR_HIP, "no HIP device available (libofdmtools_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, OTH_ERR_INVALID, "device_id out of range");
    oth_ctx *c = new (std::nothrow) oth_ctx();
    if (!c) return fail(nullptr, OTH_ERR_NOMEM, "host allocation failed");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        delete c;
        return fail(nullptr, OTH_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    }
    if (adopt) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
        c->own_stream = false;
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete c;
            return fail(nullptr, OTH_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    if (hipMalloc(&c->sink, 256) != hipSuccess || hipMalloc(&c->acc4, 4 * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->queue, 65 * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(c->queue, 0, 65 * sizeof(unsigned), c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        delete c;
        return fail(nullptr, OTH_ERR_NOMEM, "hipMalloc failed for context scratch");
    }
    c->queue_clean = true;
    c->done_count = c->queue + 64;     // zero now; every signalling finalize launch leaves it at zero again
    *out = c;
    return OTH_OK;
}

int oth_ctx_create(int device_id, oth_ctx **out) {
    OTH_TRY
    return ctx_create(device_id, nullptr, false, out);
    OTH_CATCH(nullptr)
}
int oth_ctx_create_on_stream(int device_id, void *hip_stream, oth_ctx **out) {
    OTH_TRY
    return ctx_create(device_id, hip_stream, true, out);
    OTH_CATCH(nullptr)
}

int oth_ctx_destroy(oth_ctx *c) {
    OTH_TRY
    if (!c) return OTH_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto &ev : c->events) {
        hipEventDestroy(ev.first);
        hipEventDestroy(ev.second);
    }
    for (auto &ev : c->free_events) {
        hipEventDestroy(ev.first);
        hipEventDestroy(ev.second);
    }
    for (auto &kv : c->twiddles) hipFree(kv.second);
    if (c->sink) hipFree(c->sink);
    if (c->acc4) hipFree(c->acc4);
    if (c->queue) hipFree(c->queue);
    if (c->scratch) hipFree(c->scratch);
    if (c->d_bounds) hipFree(c->d_bounds);
    if (c->h_bounds) hipHostFree(c->h_bounds);
    if (c->bounds_ev) hipEventDestroy(c->bounds_ev);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return OTH_OK;
    OTH_CATCH(nullptr)
}

const char *oth_last_error(oth_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }

int oth_ctx_sync(oth_ctx *c) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c) return fail(nullptr, OTH_ERR_INVALID, "ctx is NULL");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_ctx_device_name(oth_ctx *c, char *buf, size_t buflen) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !buf || !buflen) return fail(c, OTH_ERR_INVALID, "bad argument");
    std::snprintf(buf, buflen, "%s", c->name.c_str());
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_ctx_set_timing(oth_ctx *c, int enable) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c) return fail(nullptr, OTH_ERR_INVALID, "ctx is NULL");
    c->timing = enable != 0;
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_ctx_get_timing(oth_ctx *c, double *total_ms, uint64_t *launches, int reset) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c) return fail(nullptr, OTH_ERR_INVALID, "ctx is NULL");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &ev : c->events) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, ev.first, ev.second));
        c->total_ms += ms;
        c->free_events.push_back(ev);
    }
    c->events.clear();
    if (total_ms) *total_ms = c->total_ms;
    if (launches) *launches = c->launches;
    if (reset) {
        c->total_ms = 0.0;
        c->launches = 0;
    }
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_dev_alloc(oth_ctx *c, size_t bytes, void **dptr) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !dptr) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(c, OTH_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_dev_free(oth_ctx *c, void *dptr) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c) return fail(nullptr, OTH_ERR_INVALID, "ctx is NULL");
    if (!dptr) return OTH_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(dptr));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_memcpy_h2d(oth_ctx *c, void *dst, const void *src, size_t bytes) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !dst || !src) return fail(c, OTH_ERR_INVALID, "bad argument");
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_memcpy_d2h(oth_ctx *c, void *dst, const void *src, size_t bytes) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !dst || !src) return fail(c, OTH_ERR_INVALID, "bad argument");
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_synth_iq(oth_ctx *c, void *iq_dev, size_t nsamples, uint64_t seed, int ntones, const float *tone_amp,
                 const float *tone_freq, float dc_re, float dc_im) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !iq_dev || ntones < 0 || ntones > 8 || (ntones && (!tone_amp || !tone_freq)))
        return fail(c, OTH_ERR_INVALID, "bad argument (at most 8 tones)");
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, launch_synth((float2 *)iq_dev, nsamples, seed, ntones, tone_amp, tone_freq, dc_re, dc_im, c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_stream_read_probe(oth_ctx *c, const void *dptr, size_t bytes, int repeats, double *ms_per_pass) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !dptr || bytes < 16 || repeats == 0 || !ms_per_pass) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    // repeats < 0: the 8-bytes-per-lane variant (the access width of the FFT kernels' sample loads), |repeats| passes
    const bool narrow = repeats < 0;
    if (narrow) repeats = -repeats;
    auto probe = [&]() { return narrow ? launch_read_probe8(dptr, bytes, c->sink, c->stream) : launch_read_probe(dptr, bytes, c->sink, c->stream); };
    HIPCHK(c, probe());   // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (int i = 0; i < repeats; ++i) HIPCHK(c, probe());
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    *ms_per_pass = (double)ms / repeats;
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_iq_power(oth_ctx *c, const void *iq_dev, size_t nsamples, double *mean_re, double *mean_im, double *var) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !iq_dev || !nsamples) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, hipMemsetAsync(c->acc4, 0, 4 * sizeof(double), c->stream));
    HIPCHK(c, launch_iq_power((const float2 *)iq_dev, nsamples, c->acc4, c->stream));
    double h[4];
    HIPCHK(c, hipMemcpyAsync(h, c->acc4, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double mr = h[0] / nsamples, mi = h[1] / nsamples;
    if (mean_re) *mean_re = mr;
    if (mean_im) *mean_im = mi;
    if (var) *var = h[2] / nsamples - (mr * mr + mi * mi);
    return OTH_OK;
    OTH_CATCH(c)
}

/* ---- Welch ---------------------------------------------------------------- */

int oth_welch_plan(oth_ctx *c, int nfft, int nperseg, int noverlap, const float *window, int detrend, int scaling,
                   double fs, int fftshift, int trim_bins, oth_plan **out) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !out) return fail(c, OTH_ERR_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (nfft < 1) return fail(c, OTH_ERR_INVALID, "nfft must be positive");
    const bool any_route = !generic_supported(nfft);      // not a power of two in [64, 16384]: fft_any.hip
    if (nperseg < 1 || nperseg > nfft) return fail(c, OTH_ERR_INVALID, "need 1 <= nperseg <= nfft");
    if (noverlap < 0 || noverlap >= nperseg) return fail(c, OTH_ERR_INVALID, "need 0 <= noverlap < nperseg");
    if (detrend != OTH_DETREND_NONE && detrend != OTH_DETREND_CONSTANT && detrend != OTH_DETREND_CONSTANT_EXACT &&
        detrend != OTH_DETREND_CONSTANT_FAST)
        return fail(c, OTH_ERR_INVALID, "unknown detrend");
    const bool fast_detrend = detrend == OTH_DETREND_CONSTANT_FAST;
    if (detrend != OTH_DETREND_NONE) detrend = OTH_DETREND_CONSTANT;      // one operation: forms and builds are run_average's choice
    if (scaling < OTH_SCALE_RAW || scaling > OTH_SCALE_SPECTRUM) return fail(c, OTH_ERR_INVALID, "unknown scaling");
    if (trim_bins < 0 || 2 * trim_bins >= nfft) return fail(c, OTH_ERR_INVALID, "trim_bins out of range");
    if (!(fs > 0.0)) return fail(c, OTH_ERR_INVALID, "fs must be positive");
    if (use_device(c)) return OTH_ERR_HIP;
    oth_plan *p = new (std::nothrow) oth_plan();
    if (!p) return fail(c, OTH_ERR_NOMEM, "host allocation failed");
    p->ctx = c;
    p->nfft = nfft;
    p->nperseg = nperseg;
    p->noverlap = noverlap;
    p->step = nperseg - noverlap;
    p->detrend = detrend;
    p->fast_detrend = fast_detrend;
    p->scaling = scaling;
    p->fs = fs;
    p->fftshift = fftshift != 0;
    p->trim = trim_bins;
    if (const char *e = getenv("OTH_W4096_VARIANT")) p->tune_variant = e;      // read once, here
    if (const char *e = getenv("OTH_W4096_SCHED")) p->tune_sched = atoi(e);
    if (const char *e = getenv("OTH_W4096_CHUNK")) p->tune_chunk = atoi(e);
    if (const char *e = getenv("OTH_W4096_TAIL")) p->tune_tail = atoi(e);
    if (const char *e = getenv("OTH_PILOT_LAUNCH")) p->pilot_launch = atoi(e) != 0;
    if (const char *e = getenv("OTH_HOSTWAIT")) p->hostwait = !strcmp(e, "sync") ? 1 : 0;      // initial value of oth_plan_set_hostwait
    std::vector<float> w(nfft, 0.f);   // zero-extended so that kernels may index [0, nfft)
    double s1 = 0.0, s2 = 0.0;
    p->rect_window = true;
    for (int i = 0; i < nperseg; ++i) {
        w[i] = window ? window[i] : 1.0f;
        if (w[i] != 1.0f) p->rect_window = false;
        s1 += (double)w[i];
        s2 += (double)w[i] * (double)w[i];
    }
    switch (scaling) {
        case OTH_SCALE_DENSITY: p->scale = 1.0 / (fs * s2); break;
        case OTH_SCALE_OVER_N2: p->scale = 1.0 / ((double)nfft * (double)nfft); break;
        case OTH_SCALE_SPECTRUM: p->scale = 1.0 / (s1 * s1); break;
        default: p->scale = 1.0;
    }
    int rc = any_route ? any_tables_init(c, nfft, &p->any) : get_twiddles(c, nfft, &p->d_tw);
    if (rc) {
        delete p;
        return rc;
    }
    hipError_t e = hipMalloc(&p->d_win, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMalloc(&p->d_sum, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_win, w.data(), sizeof(float) * nfft, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_sum, 0, sizeof(float) * nfft, c->stream);
    std::vector<float> fd;
    if (e == hipSuccess && detrend == OTH_DETREND_CONSTANT &&
        ((nfft == 4096 && nperseg == 4096 && window_spectrum_table(w, fd)) ||
         (nfft == 2048 && nperseg == 2048 && window_spectrum_table_seg(w, nfft, fd)) ||
         ((nfft == 8192 || nfft == 16384) && nperseg == nfft && window_spectrum_table_16k(w, nfft, fd)))) {
        e = hipMalloc(&p->d_fd, sizeof(float) * fd.size());
        if (e == hipSuccess)
            e = hipMemcpyAsync(p->d_fd, fd.data(), sizeof(float) * fd.size(), hipMemcpyHostToDevice, c->stream);
    }
    std::vector<float> fd1x;
    if (e == hipSuccess && detrend == OTH_DETREND_CONSTANT && (nfft == 16384 || nfft == 8192) && nperseg == nfft &&
        window_spectrum_table_1x(w, nfft, fd1x)) {
        e = hipMalloc(&p->d_fd1x, sizeof(float) * fd1x.size());
        if (e == hipSuccess)
            e = hipMemcpyAsync(p->d_fd1x, fd1x.data(), sizeof(float) * fd1x.size(), hipMemcpyHostToDevice, c->stream);
    }
    std::vector<float> wpm;
    if (e == hipSuccess && any_route && nfft == 65536 && nperseg == 65536) {
        wpm.resize(65536);
        for (int n = 0; n < 32768; ++n) {
            wpm[n] = (float)((double)w[n] + (double)w[n + 32768]);
            wpm[32768 + n] = (float)((double)w[n] - (double)w[n + 32768]);
        }
        e = hipMalloc(&p->d_wpm, sizeof(float) * wpm.size());
        if (e == hipSuccess) e = hipMemcpyAsync(p->d_wpm, wpm.data(), sizeof(float) * wpm.size(), hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        oth_plan_destroy(p);
        return fail(c, OTH_ERR_HIP, std::string("plan setup: ") + hipGetErrorString(e));
    }
    *out = p;
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_plan_destroy(oth_plan *p) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return OTH_OK;
    oth_ctx *c = p->ctx;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (p->d_win) hipFree(p->d_win);
    if (p->d_wpm) hipFree(p->d_wpm);
    if (p->d_fd) hipFree(p->d_fd);
    if (p->d_fd1x) hipFree(p->d_fd1x);
    if (p->d_pilot) hipFree(p->d_pilot);
    if (p->d_partial) hipFree(p->d_partial);
    if (p->d_reduce) hipFree(p->d_reduce);
    if (p->d_out) hipFree(p->d_out);
    if (p->h_out) hipHostFree(p->h_out);
    if (p->h_seq) hipHostFree(p->h_seq);
    if (p->d_stage) hipFree(p->d_stage);
    if (p->d_sum) hipFree(p->d_sum);
    if (p->d_stream) hipFree(p->d_stream);
    any_tables_free(p->any);
    for (int i = 0; i < 4; ++i) {
        if (p->h_ring[i]) hipHostFree(p->h_ring[i]);
        if (p->h_ring_ev[i]) hipEventDestroy(p->h_ring_ev[i]);
    }
    delete p;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_set_output_db(oth_plan *p, int enable) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    p->db = enable != 0;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_set_kernel(oth_plan *p, int which) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (which < OTH_KERNEL_AUTO || which > OTH_KERNEL_TUNED) return fail(p->ctx, OTH_ERR_INVALID, "unknown kernel id");
    p->kernel = which;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_set_schedule(oth_plan *p, int which) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (which < OTH_SCHED_CONTIGUOUS || which > OTH_SCHED_DYNAMIC) return fail(p->ctx, OTH_ERR_INVALID, "unknown schedule");
    p->sched = which;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_set_tuning(oth_plan *p, const char *variant, int sched, int chunk, int tail_chunk) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (sched < -1 || sched > OTH_SCHED_DYNAMIC || chunk < 0 || tail_chunk < 0)
        return fail(p->ctx, OTH_ERR_INVALID, "bad tuning value");
    if (variant && *variant) {
        bool known = !strcmp(variant, "seg3") || !strcmp(variant, "seg4") || !strcmp(variant, "segws") ||   // 1024 / 2048
                     !strcmp(variant, "csd1") ||                                // the one-role two-channel kernel
                     !strcmp(variant, "fd") || !strcmp(variant, "td") ||        // detrend form only (run_average)
                     !strcmp(variant, "plaunch") ||                             // pilot from its own launch (run_average)
                     !strcmp(variant, "16k4") || !strcmp(variant, "16kplain") ||  // 16384 points: the 4 x 4096 build / the
                                                                                // un-pipelined one-exchange build
                     !strcmp(variant, "8kws") || !strcmp(variant, "8k1role") ||  // 8192 points, 50 % overlap: role-split / one-role
                     !strcmp(variant, "anycov") ||                              // 32768 / 65536 points: fft_any.hip's coverage kernels
                                                                                // instead of fft_tl.hip's
                     !strcmp(variant, "r16");                                   // 32768 points: fft_tl.hip's four-step route instead
                                                                                // of welch32k.hip
        for (const auto &v : kVariants) known = known || !strcmp(variant, v.tag);
        if (!known) return fail(p->ctx, OTH_ERR_UNSUPPORTED, std::string("unknown kernel build: ") + variant);
    }
    p->tune_variant = variant ? variant : "";
    p->tune_sched = sched;
    p->tune_chunk = chunk;
    p->tune_tail = tail_chunk;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_set_hostwait(oth_plan *p, int mode) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (mode != OTH_HOSTWAIT_POLL && mode != OTH_HOSTWAIT_SYNC) return fail(p->ctx, OTH_ERR_INVALID, "unknown host-wait mode");
    p->hostwait = mode;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_plan_out_len(oth_plan *p, int *n) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p || !n) return fail(p ? p->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    *n = p->nfft - 2 * p->trim;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

// averaging launch + finalize into psd_out (device memory, or a pinned host row when host_seq is given: the finalize
// launch then also publishes seq_value there once the row is complete)
static int welch_exec_dev_impl(oth_plan *p, const void *iq_dev, size_t nsamples, int nstreams, size_t stream_stride,
                               float *psd_out_dev, uint64_t *nseg_out, unsigned *host_seq, unsigned seq_value) {
    oth_ctx *c = p->ctx;
    if (!iq_dev || !psd_out_dev || nstreams < 1) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (nstreams > 1 && stream_stride < nsamples) return fail(c, OTH_ERR_INVALID, "stream_stride < nsamples");
    if (use_device(c)) return OTH_ERR_HIP;
    long long nseg = 0;
    int W = 0, layout = 0;
    int rc = run_average(p, (const float2 *)iq_dev, nullptr, nsamples, nstreams, stream_stride, &nseg, &W, &layout);
    if (rc) return rc;
    FinalizeArgs f{};
    f.partial = p->d_partial;
    f.scratch = p->d_reduce;
    f.out0 = psd_out_dev;
    f.scale = p->scale / (double)nseg;
    f.W = W;
    f.nfft = p->nfft;
    f.nch = 1;
    f.layout = layout;
    f.l1 = p->any.sh.L1, f.l2 = p->any.sh.L2;
    f.fftshift = p->fftshift;
    f.trim = p->trim;
    f.db = p->db;
    f.nout = p->nfft - 2 * p->trim;
    if (host_seq) {
        f.done_count = c->done_count;
        f.host_seq = host_seq;
        f.seq_value = seq_value;
    }
    if (int frc = finalize_and_rearm(c, f, nstreams)) return frc;
    if (nseg_out) *nseg_out = (uint64_t)nseg;
    return OTH_OK;
}

int oth_welch_exec_dev(oth_plan *p, const void *iq_dev, size_t nsamples, int nstreams, size_t stream_stride,
                       float *psd_out_dev, uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    return welch_exec_dev_impl(p, iq_dev, nsamples, nstreams, stream_stride, psd_out_dev, nseg_out, nullptr, 0u);
    OTH_CATCH((p ? p->ctx : nullptr))
}

static int stage_host(oth_plan *p, const void *x, const void *y, size_t nsamples, const float2 **dx, const float2 **dy) {
    oth_ctx *c = p->ctx;
    const size_t bytes = nsamples * sizeof(float2);
    int rc = ensure(c, &p->d_stage, &p->stage_cap, bytes * (y ? 2 : 1));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(p->d_stage, x, bytes, hipMemcpyHostToDevice, c->stream));
    *dx = p->d_stage;
    if (y) {
        HIPCHK(c, hipMemcpyAsync(p->d_stage + nsamples, y, bytes, hipMemcpyHostToDevice, c->stream));
        *dy = p->d_stage + nsamples;
    }
    return OTH_OK;
}

// ---- host-output forms: oth_welch_exec (blocking), oth_welch_exec_async / _poll / _wait (tickets) --------------------
// SURVEY 8d ends the metric at "PSD available on host".  Round 4: three launches, then hipStreamSynchronize - an
// interrupt wake-up whose latency differs by 100 us between hosts of the same pool.  Now the finalize launch writes the
// row into pinned host memory and a completion word behind it (kernels_misc.hip finalize_signal), and the host polls
// that word: first in a tight loop, then yielding the CPU between looks, and only after kPollFallbackMs through
// hipStreamSynchronize (which also turns a faulted launch into an error code instead of an endless wait).
// OTH_HOSTWAIT=sync restores the wait of round 4 for the A/B.
namespace {
constexpr double kPollSpinUs = 2000.0;         // tight polling (pause instructions only): covers a 2^28-sample launch; with
                                               // sched_yield() from 200 us on, a process with other runnable threads (bench.py
                                               // under torch) came back 30 us late (0.6256 against 0.595 ms per step)
constexpr double kPollFallbackMs = 20.0;       // then yield between looks; past this, hipStreamSynchronize (round 5: 200 ms -
                                               // a core per blocked caller for that long; OTH_HOSTWAIT_SYNC / oth_plan_set_hostwait
                                               // is the mode for flowgraphs with many blocking sensors)

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}

inline double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

// The word of a slot only ever grows (tickets t, t + 4, t + 8, ... of that slot, low 32 bits): "reached" is >= in
// wrap-safe arithmetic, so a waiter whose ticket was overtaken by a newer launch on the slot leaves its loop too (and is
// then told OTH_ERR_STATE, not that the word was never written).
inline bool seq_reached(const unsigned *word, unsigned want) {
    return (int)(__atomic_load_n(word, __ATOMIC_ACQUIRE) - want) >= 0;
}

// -> true when the word shows `want` (the row behind it is then visible to this thread)
bool poll_seq(const unsigned *word, unsigned want, double budget_ms) {
    if (seq_reached(word, want)) return true;
    const double t0 = now_us();
    for (;;) {
        for (int i = 0; i < 32; ++i) {
            if (seq_reached(word, want)) return true;
            cpu_relax();
        }
        const double dt = now_us() - t0;
        if (dt > budget_ms * 1e3) return false;
        if (dt > kPollSpinUs) sched_yield();
    }
}

int out_ring_init(oth_plan *p) {
    oth_ctx *c = p->ctx;
    if (p->h_out) return OTH_OK;
    void *rows = nullptr, *seq = nullptr;
    if (hipHostMalloc(&rows, sizeof(float) * p->nfft * oth_plan::kOutRing, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&seq, sizeof(unsigned) * 16 * oth_plan::kOutRing, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (rows) hipHostFree(rows);
        return fail(c, OTH_ERR_NOMEM, "pinned host allocation failed");
    }
    memset(seq, 0, sizeof(unsigned) * 16 * oth_plan::kOutRing);      // one word per 64-byte line
    p->h_out = (float *)rows;
    p->h_seq = (unsigned *)seq;
    return OTH_OK;
}

// Enqueue one host-output launch; the caller holds the context lock.
int welch_enqueue(oth_plan *p, const void *iq, size_t nsamples, int src_is_device, bool caller_blocks, uint64_t *ticket_out) {
    oth_ctx *c = p->ctx;
    if (!iq) return fail(c, OTH_ERR_INVALID, "iq is NULL");
    if (nsamples < (size_t)p->nperseg) return fail(c, OTH_ERR_INVALID, "input shorter than nperseg");
    if (use_device(c)) return OTH_ERR_HIP;
    int rc = out_ring_init(p);
    if (rc) return rc;
    const uint64_t ticket = p->next_out_ticket;
    const int slot = (int)(ticket % oth_plan::kOutRing);
    unsigned *word = p->h_seq + 16 * slot;
    // the slot's previous launch (kOutRing tickets ago) must have delivered before its row is written again
    if (p->out_ticket[slot] && !seq_reached(word, (unsigned)p->out_ticket[slot])) {
        if (!poll_seq(word, (unsigned)p->out_ticket[slot], kPollFallbackMs)) HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const float2 *dx = (const float2 *)iq, *dy = nullptr;
    if (!src_is_device) {
        // the caller's buffer is valid during the call only (sync_block.work()): pageable memory is staged by the
        // runtime before hipMemcpyAsync returns; a pinned / registered source would be read asynchronously, so that
        // copy is awaited (the one case in which the asynchronous form waits for the stream)
        const size_t bytes = nsamples * sizeof(float2);
        if (!caller_blocks && bytes <= kPinnedStageMax) {
            // work()-sized buffers: through a pinned slot (as oth_welch_accumulate), so that the call returns at once -
            // from pageable memory hipMemcpyAsync may hold the host until the stream has drained
            if ((rc = ensure(c, &p->d_stage, &p->stage_cap, bytes))) return rc;
            const unsigned rs = p->h_ring_next++ & 3u;
            if (!p->h_ring_ev[rs]) HIPCHK(c, hipEventCreateWithFlags(&p->h_ring_ev[rs], hipEventDisableTiming));
            else HIPCHK(c, hipEventSynchronize(p->h_ring_ev[rs]));
            if (p->h_ring_cap[rs] < bytes) {
                if (p->h_ring[rs]) HIPCHK(c, hipHostFree(p->h_ring[rs]));
                p->h_ring[rs] = nullptr;
                p->h_ring_cap[rs] = 0;
                HIPCHK(c, hipHostMalloc(&p->h_ring[rs], bytes + bytes / 2 + 4096, hipHostMallocDefault));
                p->h_ring_cap[rs] = bytes + bytes / 2 + 4096;
            }
            memcpy(p->h_ring[rs], iq, bytes);
            HIPCHK(c, hipMemcpyAsync(p->d_stage, p->h_ring[rs], bytes, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(p->h_ring_ev[rs], c->stream));
            dx = p->d_stage;
        } else if (!caller_blocks && host_ptr_is_pinned(iq)) {
            if ((rc = ensure(c, &p->d_stage, &p->stage_cap, bytes))) return rc;
            if ((rc = copy_in_and_wait(c, p->d_stage, iq, bytes))) return rc;
            dx = p->d_stage;
        } else if ((rc = stage_host(p, iq, nullptr, nsamples, &dx, &dy))) {
            return rc;
        }
    }
    uint64_t nseg = 0;
    if ((rc = welch_exec_dev_impl(p, dx, nsamples, 1, nsamples, p->h_out + (size_t)slot * p->nfft, &nseg, word,
                                  (unsigned)ticket)))
        return rc;
    p->out_ticket[slot] = ticket;
    p->out_nseg[slot] = nseg;
    p->next_out_ticket = ticket + 1;
    *ticket_out = ticket;
    return OTH_OK;
}

// Collect a ticket: wait == 0 looks once, wait == 1 polls (outside the context lock) and falls back to the stream.
int welch_collect(oth_plan *p, uint64_t ticket, float *psd_out, uint64_t *nseg_out, int *ready, bool wait) {
    oth_ctx *c = p->ctx;
    const int slot = (int)(ticket % oth_plan::kOutRing);
    const unsigned *word;
    uint64_t nseg;
    {
        CtxGuard guard_(c);
        if (!ticket || !p->h_out || p->out_ticket[slot] != ticket)
            return fail(c, OTH_ERR_STATE, "ticket unknown or overwritten (the ring keeps the last 4 launches)");
        word = p->h_seq + 16 * slot;
        nseg = p->out_nseg[slot];
    }
    bool done = seq_reached(word, (unsigned)ticket);
    if (!done && wait) {
        if (p->hostwait == 0) done = poll_seq(word, (unsigned)ticket, kPollFallbackMs);
        if (!done) {
            CtxGuard guard_(c);
            if (use_device(c)) return OTH_ERR_HIP;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            done = seq_reached(word, (unsigned)ticket);
            if (!done) return fail(c, OTH_ERR_INTERNAL, "stream idle but the completion word was never written");
        }
    }
    if (ready) *ready = done ? 1 : 0;
    if (!done) return OTH_OK;
    {
        CtxGuard guard_(c);      // (a newer launch may have taken the slot while this thread was polling)
        if (p->out_ticket[slot] != ticket)
            return fail(c, OTH_ERR_STATE, "ticket overwritten while waiting (the ring keeps the last 4 launches)");
        if (psd_out) memcpy(psd_out, p->h_out + (size_t)slot * p->nfft, sizeof(float) * (p->nfft - 2 * p->trim));
    }
    if (nseg_out) *nseg_out = nseg;
    return OTH_OK;
}
}  // namespace

int oth_welch_exec_async(oth_plan *p, const void *iq, size_t nsamples, int src_is_device, uint64_t *ticket_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (!ticket_out) return fail(p->ctx, OTH_ERR_INVALID, "ticket_out is NULL");
    *ticket_out = 0;
    return welch_enqueue(p, iq, nsamples, src_is_device, false, ticket_out);
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_poll(oth_plan *p, uint64_t ticket, float *psd_out, uint64_t *nseg_out, int *ready) {
    OTH_TRY
    if (!p || !ready) return fail(p ? p->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    *ready = 0;
    return welch_collect(p, ticket, psd_out, nseg_out, ready, false);
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_wait(oth_plan *p, uint64_t ticket, float *psd_out, uint64_t *nseg_out) {
    OTH_TRY
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    return welch_collect(p, ticket, psd_out, nseg_out, nullptr, true);
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_exec(oth_plan *p, const void *iq, size_t nsamples, int src_is_device, float *psd_out,
                   uint64_t *nseg_out) {
    OTH_TRY
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    if (!iq || !psd_out) return fail(p->ctx, OTH_ERR_INVALID, "bad argument");
    uint64_t ticket = 0;
    std::lock_guard<std::mutex> one_at_a_time(p->exec_mu);      // blocking callers of ONE plan, any number of threads
    {
        CtxGuard guard_(p->ctx);
        if (int rc = welch_enqueue(p, iq, nsamples, src_is_device, true, &ticket)) return rc;
    }
    return welch_collect(p, ticket, psd_out, nseg_out, nullptr, true);      // polls outside the context lock
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_partial_dev(oth_plan *p, const void *iq_dev, size_t nsamples, float *sum_out_dev, uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!iq_dev || !sum_out_dev) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    long long nseg = 0;
    int W = 0, layout = 0;
    int rc = run_average(p, (const float2 *)iq_dev, nullptr, nsamples, 1, nsamples, &nseg, &W, &layout);
    if (rc) return rc;
    FinalizeArgs f{};
    f.partial = p->d_partial;
    f.scratch = p->d_reduce;
    f.out0 = sum_out_dev;
    f.scale = 1.0;
    f.W = W;
    f.nfft = p->nfft;
    f.nch = 1;
    f.layout = layout;
    f.l1 = p->any.sh.L1, f.l2 = p->any.sh.L2;
    f.nout = p->nfft;
    if (int frc = finalize_and_rearm(c, f, 1)) return frc;
    if (nseg_out) *nseg_out = (uint64_t)nseg;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_scale_dev(oth_plan *p, const float *sum_dev, uint64_t nseg_total, float *psd_out_dev) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!sum_dev || !psd_out_dev || !nseg_total) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, launch_scale(sum_dev, psd_out_dev, p->nfft, p->scale / (double)nseg_total, p->fftshift, p->trim, p->db,
                           c->stream));
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_reset(oth_plan *p) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, hipMemsetAsync(p->d_sum, 0, sizeof(float) * p->nfft, c->stream));
    p->nseg_total = 0;
    p->carry = 0;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_accumulate(oth_plan *p, const void *iq_host, size_t nsamples) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!iq_host && nsamples) return fail(c, OTH_ERR_INVALID, "iq is NULL");
    if (!nsamples) return OTH_OK;
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t total = p->carry + nsamples;
    int rc = ensure_keep(c, &p->d_stream, &p->stream_cap, total * sizeof(float2), p->carry * sizeof(float2));
    if (rc) return rc;
    const bool pinned_src = nsamples * sizeof(float2) > kPinnedStageMax && host_ptr_is_pinned(iq_host);
    if (pinned_src && nsamples * sizeof(float2) > kPinnedRingMax) {
        if ((rc = copy_in_and_wait(c, p->d_stream + p->carry, iq_host, nsamples * sizeof(float2)))) return rc;
    } else if (nsamples * sizeof(float2) > kPinnedStageMax && !pinned_src) {
        // large chunks: the runtime's own staged copy from pageable memory is faster than a host memcpy into a pinned
        // slot (55 against 33 GB/s at 32 MiB); it returns once the caller's buffer has been read.  (A pinned /
        // registered source would be read asynchronously: it takes the ring below whatever its size.)
        HIPCHK(c, hipMemcpyAsync(p->d_stream + p->carry, iq_host, nsamples * sizeof(float2), hipMemcpyHostToDevice,
                                 c->stream));
    } else {
        // the caller's buffer is only valid during the call (sync_block.work contract): it is copied into a pinned
        // slot, the H2D copy is enqueued from there and the call returns without waiting for the GPU (a slot is
        // reused four calls later; only then, if the GPU is still that far behind, does the call wait)
        const unsigned slot = p->h_ring_next++ & 3u;
        const size_t bytes = nsamples * sizeof(float2);
        if (!p->h_ring_ev[slot]) HIPCHK(c, hipEventCreateWithFlags(&p->h_ring_ev[slot], hipEventDisableTiming));
        else HIPCHK(c, hipEventSynchronize(p->h_ring_ev[slot]));
        if (p->h_ring_cap[slot] < bytes) {
            if (p->h_ring[slot]) HIPCHK(c, hipHostFree(p->h_ring[slot]));
            p->h_ring[slot] = nullptr;
            p->h_ring_cap[slot] = 0;
            HIPCHK(c, hipHostMalloc(&p->h_ring[slot], bytes + bytes / 2 + 4096, hipHostMallocDefault));
            p->h_ring_cap[slot] = bytes + bytes / 2 + 4096;
        }
        memcpy(p->h_ring[slot], iq_host, bytes);
        HIPCHK(c, hipMemcpyAsync(p->d_stream + p->carry, p->h_ring[slot], bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(p->h_ring_ev[slot], c->stream));
    }
    if (total < (size_t)p->nperseg) {
        p->carry = total;
        return OTH_OK;
    }
    long long nseg = 0;
    int W = 0, layout = 0;
    if ((rc = run_average(p, p->d_stream, nullptr, total, 1, total, &nseg, &W, &layout))) return rc;
    FinalizeArgs f{};
    f.partial = p->d_partial;
    f.scratch = p->d_reduce;
    f.out0 = p->d_sum;
    f.scale = 1.0;
    f.W = W;
    f.nfft = p->nfft;
    f.nch = 1;
    f.layout = layout;
    f.l1 = p->any.sh.L1, f.l2 = p->any.sh.L2;
    f.nout = p->nfft;
    f.accumulate = 1;
    if (int frc = finalize_and_rearm(c, f, 1)) return frc;
    p->nseg_total += (uint64_t)nseg;
    // keep the samples the next segment still needs
    const size_t consumed = (size_t)nseg * (size_t)p->step;
    const size_t keep = total - consumed;
    if (keep) {
        // regions may overlap when keep > consumed: bounce through the partial-free tail of d_stage
        if (keep <= consumed) {
            HIPCHK(c, hipMemcpyAsync(p->d_stream, p->d_stream + consumed, keep * sizeof(float2),
                                     hipMemcpyDeviceToDevice, c->stream));
        } else {
            if ((rc = ensure(c, &p->d_stage, &p->stage_cap, keep * sizeof(float2)))) return rc;
            HIPCHK(c, hipMemcpyAsync(p->d_stage, p->d_stream + consumed, keep * sizeof(float2),
                                     hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(p->d_stream, p->d_stage, keep * sizeof(float2), hipMemcpyDeviceToDevice,
                                     c->stream));
        }
    }
    p->carry = keep;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_welch_finalize(oth_plan *p, float *psd_out, uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!psd_out) return fail(c, OTH_ERR_INVALID, "psd_out is NULL");
    if (!p->nseg_total) return fail(c, OTH_ERR_STATE, "no complete segment accumulated yet");
    if (use_device(c)) return OTH_ERR_HIP;
    int rc = ensure(c, &p->d_out, &p->out_cap, sizeof(float) * 5 * p->nfft);
    if (rc) return rc;
    const int nout = p->nfft - 2 * p->trim;
    HIPCHK(c, launch_scale(p->d_sum, p->d_out, p->nfft, p->scale / (double)p->nseg_total, p->fftshift, p->trim,
                           p->db, c->stream));
    HIPCHK(c, hipMemcpyAsync(psd_out, p->d_out, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nseg_out) *nseg_out = p->nseg_total;
    return oth_welch_reset(p);
    OTH_CATCH((p ? p->ctx : nullptr))
}

// Averaging launch + cross-workgroup reduction of the two-channel path.  raw: unscaled sums in natural
// order (no shift / trim), the time-sharded form; else the plan's scale, shift and trim.
static int csd_run(oth_plan *p, const float2 *dx, const float2 *dy, size_t nsamples, bool raw, float *o_xx,
                   float *o_yy, float *o_xy, float *o_c, uint64_t *nseg_out) {
    oth_ctx *c = p->ctx;
    long long nseg = 0;
    int W = 0, layout = 0;
    int rc = run_average(p, dx, dy, nsamples, 1, nsamples, &nseg, &W, &layout);
    if (rc) return rc;
    FinalizeArgs f{};
    f.partial = p->d_partial;
    f.scratch = p->d_reduce;
    f.out0 = o_xx;
    f.out1 = o_yy;
    f.out2 = o_xy;
    f.out3 = o_c;
    f.scale = raw ? 1.0 : p->scale / (double)nseg;
    f.W = W;
    f.nfft = p->nfft;
    f.nch = 4;
    f.layout = layout;
    f.l1 = p->any.sh.L1, f.l2 = p->any.sh.L2;
    f.fftshift = raw ? 0 : p->fftshift;
    f.trim = raw ? 0 : p->trim;
    f.nout = raw ? p->nfft : p->nfft - 2 * p->trim;
    if (int frc = finalize_and_rearm(c, f, 1)) return frc;
    if (nseg_out) *nseg_out = (uint64_t)nseg;
    return OTH_OK;
}

int oth_csd_exec_dev(oth_plan *p, const void *x_dev, const void *y_dev, size_t nsamples, float *pxx_dev,
                     float *pyy_dev, float *pxy_dev, float *cxy_dev, uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!x_dev || !y_dev) return fail(c, OTH_ERR_INVALID, "x/y is NULL");
    if (p->db) return fail(c, OTH_ERR_UNSUPPORTED, "dB output is not defined for the cross spectrum");
    if (nsamples < (size_t)p->nperseg) return fail(c, OTH_ERR_INVALID, "input shorter than nperseg");
    if (use_device(c)) return OTH_ERR_HIP;
    return csd_run(p, (const float2 *)x_dev, (const float2 *)y_dev, nsamples, false, pxx_dev, pyy_dev, pxy_dev,
                   cxy_dev, nseg_out);
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_csd_partial_dev(oth_plan *p, const void *x_dev, const void *y_dev, size_t nsamples, float *sums_out_dev,
                        uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!x_dev || !y_dev || !sums_out_dev) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (nsamples < (size_t)p->nperseg) return fail(c, OTH_ERR_INVALID, "input shorter than nperseg");
    if (use_device(c)) return OTH_ERR_HIP;
    const int N = p->nfft;
    return csd_run(p, (const float2 *)x_dev, (const float2 *)y_dev, nsamples, true, sums_out_dev, sums_out_dev + N,
                   sums_out_dev + 2 * N, nullptr, nseg_out);
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_csd_scale_dev(oth_plan *p, const float *sums_dev, uint64_t nseg_total, float *pxx_dev, float *pyy_dev,
                      float *pxy_dev, float *cxy_dev) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!sums_dev || !nseg_total) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, launch_csd_scale(sums_dev, p->nfft, p->scale / (double)nseg_total, p->fftshift, p->trim, pxx_dev,
                               pyy_dev, pxy_dev, cxy_dev, c->stream));
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_csd_exec(oth_plan *p, const void *x, const void *y, size_t nsamples, int src_is_device, float *pxx,
                 float *pyy, float *pxy, float *cxy, uint64_t *nseg_out) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p) return fail(nullptr, OTH_ERR_INVALID, "plan is NULL");
    oth_ctx *c = p->ctx;
    if (!x || !y) return fail(c, OTH_ERR_INVALID, "x/y is NULL");
    if (p->db) return fail(c, OTH_ERR_UNSUPPORTED, "dB output is not defined for the cross spectrum");
    if (nsamples < (size_t)p->nperseg) return fail(c, OTH_ERR_INVALID, "input shorter than nperseg");
    if (use_device(c)) return OTH_ERR_HIP;
    const float2 *dx = (const float2 *)x, *dy = (const float2 *)y;
    int rc;
    if (!src_is_device && (rc = stage_host(p, x, y, nsamples, &dx, &dy))) return rc;
    if ((rc = ensure(c, &p->d_out, &p->out_cap, sizeof(float) * 5 * p->nfft))) return rc;
    const int nout = p->nfft - 2 * p->trim;
    float *o0 = p->d_out, *o1 = p->d_out + p->nfft, *o2 = p->d_out + 2 * p->nfft, *o3 = p->d_out + 4 * p->nfft;
    if ((rc = csd_run(p, dx, dy, nsamples, false, o0, o1, o2, o3, nseg_out))) return rc;
    if (pxx) HIPCHK(c, hipMemcpyAsync(pxx, o0, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
    if (pyy) HIPCHK(c, hipMemcpyAsync(pyy, o1, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
    if (pxy) HIPCHK(c, hipMemcpyAsync(pxy, o2, sizeof(float) * 2 * nout, hipMemcpyDeviceToHost, c->stream));
    if (cxy) HIPCHK(c, hipMemcpyAsync(cxy, o3, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

/* ---- periodogram chain ------------------------------------------------------ */

int oth_chain_create(oth_ctx *c, int nfft, const float *window, int fftshift, int epilogue, int keep_one_in_n,
                     oth_chain **out) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !out) return fail(c, OTH_ERR_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (nfft < 1) return fail(c, OTH_ERR_INVALID, "nfft must be positive");
    const bool any_route = !generic_supported(nfft);      // not a power of two in [64, 16384]: fft_any.hip
    if (epilogue < OTH_EPI_MAG || epilogue > OTH_EPI_MAG2_OVER_N2) return fail(c, OTH_ERR_INVALID, "unknown epilogue");
    if (keep_one_in_n < 1) return fail(c, OTH_ERR_INVALID, "keep_one_in_n must be >= 1");
    if (use_device(c)) return OTH_ERR_HIP;
    oth_chain *h = new (std::nothrow) oth_chain();
    if (!h) return fail(c, OTH_ERR_NOMEM, "host allocation failed");
    h->ctx = c;
    h->nfft = nfft;
    h->fftshift = fftshift != 0;
    h->epilogue = epilogue;
    h->keep_n = h->count = keep_one_in_n;
    int rc = any_route ? any_tables_init(c, nfft, &h->any) : get_twiddles(c, nfft, &h->d_tw);
    if (rc) {
        delete h;
        return rc;
    }
    std::vector<float> w(nfft);
    h->rect = true;
    for (int i = 0; i < nfft; ++i) {
        w[i] = window ? window[i] : 1.0f;
        if (w[i] != 1.0f) h->rect = false;
    }
    hipError_t e = hipMalloc(&h->d_win, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMalloc(&h->d_iir, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMalloc(&h->d_peak, sizeof(float) * nfft);
    if (e == hipSuccess) e = hipMalloc(&h->d_peak_init, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_win, w.data(), sizeof(float) * nfft, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_iir, 0, sizeof(float) * nfft, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_peak, 0, sizeof(float) * nfft, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_peak_init, 0, sizeof(int), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        oth_chain_destroy(h);
        return fail(c, OTH_ERR_HIP, std::string("chain setup: ") + hipGetErrorString(e));
    }
    *out = h;
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_chain_destroy(oth_chain *h) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return OTH_OK;
    oth_ctx *c = h->ctx;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (h->d_win) hipFree(h->d_win);
    if (h->d_buf) hipFree(h->d_buf);
    if (h->d_rows) hipFree(h->d_rows);
    if (h->d_iir) hipFree(h->d_iir);
    if (h->d_peak) hipFree(h->d_peak);
    if (h->d_peak_init) hipFree(h->d_peak_init);
    if (h->d_stage) hipFree(h->d_stage);
    if (h->d_partial) hipFree(h->d_partial);
    if (h->d_tail) hipFree(h->d_tail);
    if (h->d_out) hipFree(h->d_out);
    any_tables_free(h->any);
    for (int i = 0; i < oth_chain::kRing; ++i) {
        if (h->h_in[i]) hipHostFree(h->h_in[i]);
        if (h->h_row[i]) hipHostFree(h->h_row[i]);
        if (h->ev[i]) hipEventDestroy(h->ev[i]);
    }
    delete h;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_set_keep_one_in_n(oth_chain *h, int n) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    if (n < 1) return fail(h->ctx, OTH_ERR_INVALID, "keep_one_in_n must be >= 1");
    h->keep_n = h->count = n;   // keep_one_in_n::set_n restarts the count (a partial vector whose samples were skipped as
                                // dropped stays unemitted: chain_feed's leftover_stale)
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_set_iir_log(oth_chain *h, float alpha, float k_db) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    h->do_iir = alpha > 0.f;
    h->alpha = alpha;
    h->kdb = k_db;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_set_peak_hold(oth_chain *h, int enable) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    h->do_peak = enable != 0;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_set_kernel(oth_chain *h, int which) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    if (which < OTH_KERNEL_AUTO || which > OTH_KERNEL_TUNED) return fail(h->ctx, OTH_ERR_INVALID, "unknown kernel id");
    h->kernel = which;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_reset(oth_chain *h) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    oth_ctx *c = h->ctx;
    if (use_device(c)) return OTH_ERR_HIP;
    HIPCHK(c, hipMemsetAsync(h->d_iir, 0, sizeof(float) * h->nfft, c->stream));
    HIPCHK(c, hipMemsetAsync(h->d_peak, 0, sizeof(float) * h->nfft, c->stream));
    HIPCHK(c, hipMemsetAsync(h->d_peak_init, 0, sizeof(int), c->stream));
    h->peak_flag_set = false;
    h->leftover = 0;
    h->leftover_stale = false;
    h->count = h->keep_n;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

// The fused path (segfft.hip): FFT + epilogue + IIR / peak accumulation in one launch over all kept vectors, a
// small tail kernel for the state and the rows the caller wants.  Covers nfft 1024 / 2048 / 4096 with at most one
// of {IIR + log, peak hold} and up to kTailRows rows handed back.
constexpr long long kTailRows = 256;

static bool chain_fused_ok(const oth_chain *h, long long give) {
    const bool big = h->nfft == 8192 || h->nfft == 16384;      // welch16k.hip's chain build
    if (h->kernel == OTH_KERNEL_GENERIC || !(seg_supported(h->nfft) || big)) return false;
    if (h->do_iir && h->do_peak) return false;
    if (h->do_iir && !(h->alpha > 0.f && h->alpha <= 1.f)) return false;
    return give <= kTailRows;
}

static int chain_launch_fused(oth_chain *h, const float2 *x, long long first_vec, long long nrows, float *rows_last,
                              long long give) {
    oth_ctx *c = h->ctx;
    const int N = h->nfft;
    SegArgs a{};
    a.x = x;
    a.stream_stride = 0;
    a.nstreams = 1;
    a.win = h->d_win;
    a.tw = h->d_tw;
    a.step = (long long)h->keep_n * N;
    a.first = first_vec * N;
    a.nseg = nrows;
    a.detrend = 0;
    a.chain = 1;
    a.epilogue = h->epilogue;
    a.scale = h->epilogue == OTH_EPI_MAG2_OVER_N2 ? (float)(1.0 / ((double)N * (double)N)) : 1.0f;
    a.fftshift = h->fftshift;
    a.store_from = nrows - give;
    int rc;
    if (h->do_iir) {
        a.acc_mode = 1;
        a.acc_end = a.store_from;
        a.l2 = h->alpha >= 1.f ? -INFINITY : log2f(1.0f - h->alpha);
        if (give) {
            if ((rc = ensure(c, &h->d_rows, &h->rows_cap, sizeof(float) * (size_t)give * N))) return rc;
            a.rows = h->d_rows;      // raw |X|^2 rows; the tail kernel turns them into dB rows
        }
    } else if (h->do_peak) {
        a.acc_mode = 2;
        a.acc_end = nrows;
        a.rows = rows_last;
    } else {
        // no state: rows nobody asked for are not computed at all (latest wins)
        if (!give) return OTH_OK;
        a.acc_mode = 3;
        a.acc_end = 0;
        a.first += a.store_from * a.step;
        a.nseg = give;
        a.store_from = 0;
        a.rows = rows_last;
    }
    const bool big = N >= 8192;      // one workgroup per segment: 2 (8192) / 1 (16384) per CU
    const int tpc = big ? (N == 8192 ? 2 : 1) : seg_teams_per_cu(N, 2, false);
    // segments per chunk of the interleaved schedule: 8 once every team gets two chunks (+2-4 % over 4), fewer for
    // short pushes so that more teams take part
    const long long teams_max = (long long)c->cu_count * tpc;
    a.chunk = a.nseg >= 16 * teams_max ? 8 : (a.nseg >= 4 * teams_max ? 4 : 2);
    long long nchunks = (a.nseg + a.chunk - 1) / a.chunk;
    long long W = (long long)c->cu_count * tpc;
    if (W > nchunks) W = nchunks;
    if (W < 1) W = 1;
    a.wg_per_stream = (int)W;
    a.sched = 1;                     // interleaved chunks: chunk c = team, team + W, ...
    a.tail_chunk = a.chunk;
    a.nbig = a.nseg / a.chunk;
    a.queue = nullptr;
    int groups = 0;
    if (a.acc_mode != 3) {
        if ((rc = ensure(c, &h->d_partial, &h->partial_cap, sizeof(float) * (size_t)W * N))) return rc;
        a.partial = h->d_partial;
        groups = chain_tail_groups((int)W, N);
        if (groups && (rc = ensure(c, &h->d_tail, &h->tail_cap, sizeof(float) * (size_t)groups * N))) return rc;
    }
    // 16384 points: the one-exchange pipelined loop (welch16k1x.hip, round 4); OTH_CHAIN16K=old keeps the 4 x 4096 build
    // (A/B).  At 8192 points the chain stays on the 2 x 4096 build: the 8-wave one-exchange loop with the chain's epilogue
    // spills 20 registers and measured 48.8-49.3 % against 53.4-54.1 % (windowed), 52.5 against 53.0 % (rectangular) on
    // the same box (round 5, tools/archive/ab_8k.sh; OTH_CHAIN16K=x1 selects it for the A/B)
    static const char *chain16k_mode = getenv("OTH_CHAIN16K");
    const bool x1 = (N == 16384 && !(chain16k_mode && !strcmp(chain16k_mode, "old"))) ||
                    (N == 8192 && chain16k_mode && !strcmp(chain16k_mode, "x1"));
    {
        Timed tm(c);      // the whole push: transform kernel + cross-team reduction + state / rows
        HIPCHK(c, big ? (x1 ? launch_chain16k1x(N, a, h->rect, c->stream) : launch_chain16k(N, a, h->rect, c->stream))
                      : launch_seg(N, a, 2, false, c->stream));
        h->ops += 1;
        if (a.acc_mode != 3) h->ops += (groups ? 2 : 1) + ((a.acc_mode == 1 && give > 8) ? 1 : 0);      // [reduce +] state [+ rows]
        if (a.acc_mode != 3)
            HIPCHK(c, launch_chain_tail(h->d_partial, groups ? h->d_tail : nullptr, (int)W, N, big ? (x1 ? (N == 16384 ? 4 : 5) : (N == 16384 ? 2 : 3)) : 0,
                                        h->fftshift, a.acc_mode, a.acc_end,
                                        h->alpha, h->kdb, h->d_iir, h->d_peak, h->d_rows, h->do_iir ? give : 0, rows_last,
                                        c->stream));
    }
    if (a.acc_mode == 2 && !h->peak_flag_set) {      // the coverage path (rows_epilogue_kernel) reads the flag
        HIPCHK(c, launch_set_flag(h->d_peak_init, 1, c->stream));
        h->ops += 1;
        h->peak_flag_set = true;
    }
    return OTH_OK;
}

// `nrows` kept vectors of x, vector index first_vec + r * keep_n (r = 0 .. nrows-1), through FFT + epilogue in
// time order (IIR / peak state advance); the LAST `give` post-epilogue rows land in rows_last (device).
static int chain_launch(oth_chain *h, const float2 *x, long long first_vec, long long nrows, float *rows_last,
                        long long give) {
    oth_ctx *c = h->ctx;
    const int N = h->nfft;
    if (!rows_last) give = 0;
    if (chain_fused_ok(h, give)) return chain_launch_fused(h, x, first_vec, nrows, rows_last, give);
    if (!h->do_iir && !h->do_peak) {      // no state: rows nobody asked for are not computed at all (latest wins)
        if (!give) return OTH_OK;
        first_vec += (nrows - give) * h->keep_n;
        nrows = give;
    }
    int rc = ensure(c, &h->d_rows, &h->rows_cap, sizeof(float) * (size_t)nrows * N);
    if (rc) return rc;
    PgramArgs a;
    a.x = x;
    a.win = h->d_win;
    a.tw = h->d_tw;
    a.rows = h->d_rows;
    a.first_vec = first_vec;
    a.nrows = nrows;
    a.keep_n = h->keep_n;
    a.fftshift = h->fftshift;
    a.epilogue = h->epilogue;
    a.scale = h->epilogue == OTH_EPI_MAG2_OVER_N2 ? (float)(1.0 / ((double)N * (double)N)) : 1.0f;
    if (h->any.sh.kind != ANY_NONE) {      // lengths outside the power-of-two kernels (fft_any.hip)
        Timed tm(c);
        if ((rc = any_run(c, h->any, x, nullptr, first_vec * N, (long long)h->keep_n * N, N, h->d_win, false, nrows, nullptr, 0,
                          h->d_rows, a.epilogue, a.scale, a.fftshift)))
            return rc;
    } else {
        Timed tm(c);
        HIPCHK(c, launch_pgram(N, a, c->stream));
    }
    h->ops += h->any.sh.kind == ANY_NONE ? 1 : 3;      // (the any-length routes: one to three launches per chunk)
    if (h->do_iir || h->do_peak) {
        HIPCHK(c, launch_rows_epilogue(h->d_rows, nrows, N, h->alpha, h->kdb, h->d_iir, h->d_peak, h->d_peak_init,
                                       h->do_iir, h->do_peak, c->stream));
        h->ops += h->do_peak ? 2 : 1;
    }
    if (h->do_peak && nrows > 0) h->peak_flag_set = true;
    if (rows_last && give > 0) {      // (rows_last may be pinned host memory - the asynchronous work() form: hipMemcpyDefault)
        HIPCHK(c, hipMemcpyAsync(rows_last, h->d_rows + (size_t)(nrows - give) * N, sizeof(float) * (size_t)give * N,
                                 hipMemcpyDefault, c->stream));
        h->ops += 1;
    }
    return OTH_OK;
}

// Feed nsamples device-resident samples: completes the vector left over from the previous call, runs the full
// vectors straight from `src`, keeps the incomplete tail.  Asynchronous on the context's stream.  rows_dev (may
// be NULL) receives the last min(rows, capacity) rows of this call in time order.
static int chain_feed(oth_chain *h, const float2 *src, size_t nsamples, float *rows_dev, size_t capacity,
                      uint64_t *nrows_out) {
    oth_ctx *c = h->ctx;
    const int N = h->nfft;
    int rc = ensure(c, &h->d_buf, &h->buf_cap, sizeof(float2) * (size_t)N);
    if (rc) return rc;
    bool head = false;           // a vector completed in d_buf
    if (h->leftover) {
        const size_t take = nsamples < (size_t)N - h->leftover ? nsamples : (size_t)N - h->leftover;
        HIPCHK(c, hipMemcpyAsync(h->d_buf + h->leftover, src, take * sizeof(float2), hipMemcpyDeviceToDevice,
                                 c->stream));
        h->ops += 1;
        h->leftover += take;
        src += take;
        nsamples -= take;
        if (h->leftover == (size_t)N) {
            head = true;
            h->leftover = 0;
        }
    }
    const long long nvec = (long long)(nsamples / N);
    // keep_one_in_n: `count` vectors to go until the next kept one (GNU Radio keeps the LAST of every n)
    long long k_head = 0;
    if (head) {
        if (--h->count == 0) {
            k_head = h->leftover_stale ? 0 : 1;      // (stale: its first samples were never copied - see chain_push_dropped)
            h->count = h->keep_n;
        }
        h->leftover_stale = false;
    }
    long long k_body = 0, first = h->count - 1;
    if (nvec > first) k_body = 1 + (nvec - 1 - first) / h->keep_n;
    if (k_body == 0) {
        h->count -= (int)nvec;
    } else {
        const long long last = first + (k_body - 1) * h->keep_n;
        h->count = h->keep_n - (int)(nvec - 1 - last);
    }
    const long long give_body = k_body < (long long)capacity ? k_body : (long long)capacity;
    const long long give_head = k_head < (long long)capacity - give_body ? k_head : (long long)capacity - give_body;
    if (k_head && (rc = chain_launch(h, h->d_buf, 0, 1, rows_dev, rows_dev ? give_head : 0))) return rc;
    if (k_body && (rc = chain_launch(h, src, first, k_body, rows_dev ? rows_dev + (size_t)give_head * N : nullptr,
                                     rows_dev ? give_body : 0)))
        return rc;
    const size_t used = (size_t)nvec * N, keep = nsamples - used;
    if (keep) {      // d_buf is free again: a completed head vector has been consumed by the launch above (stream order)
        HIPCHK(c, hipMemcpyAsync(h->d_buf, src + used, keep * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
        h->ops += 1;
        h->leftover = keep;
        h->leftover_stale = false;
    }
    if (nrows_out) *nrows_out = (uint64_t)(k_head + k_body);
    return OTH_OK;
}

int oth_chain_push_dev(oth_chain *h, const void *iq_dev, size_t nsamples, float *rows_out_dev, size_t rows_capacity,
                       uint64_t *nrows_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    oth_ctx *c = h->ctx;
    if (nrows_out) *nrows_out = 0;
    if (!iq_dev && nsamples) return fail(c, OTH_ERR_INVALID, "iq is NULL");
    if (!nsamples) return OTH_OK;
    if (use_device(c)) return OTH_ERR_HIP;
    h->ops = 0;
    return chain_feed(h, (const float2 *)iq_dev, nsamples, rows_out_dev, rows_out_dev ? rows_capacity : 0, nrows_out);
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_push(oth_chain *h, const void *iq, size_t nsamples, int src_is_device, float *rows_out,
                   size_t rows_capacity, uint64_t *nrows_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    oth_ctx *c = h->ctx;
    if (nrows_out) *nrows_out = 0;
    if (!iq && nsamples) return fail(c, OTH_ERR_INVALID, "iq is NULL");
    if (!nsamples) return OTH_OK;
    if (use_device(c)) return OTH_ERR_HIP;
    const int N = h->nfft;
    const float2 *src = (const float2 *)iq;
    int rc;
    h->ops = 0;
    if (!src_is_device) {
        if ((rc = ensure(c, &h->d_stage, &h->stage_cap, nsamples * sizeof(float2)))) return rc;
        HIPCHK(c, hipMemcpyAsync(h->d_stage, iq, nsamples * sizeof(float2), hipMemcpyHostToDevice, c->stream));
        src = h->d_stage;
    }
    // rows this call can produce at most: one completed leftover vector + the full vectors of the new samples
    size_t cap = rows_out ? rows_capacity : 0;
    const size_t most = nsamples / N + 2;
    if (cap > most) cap = most;
    if (cap && (rc = ensure(c, &h->d_out, &h->out_cap, sizeof(float) * cap * N))) return rc;
    uint64_t nrows = 0;
    if ((rc = chain_feed(h, src, nsamples, cap ? h->d_out : nullptr, cap, &nrows))) return rc;
    const size_t give = nrows < cap ? (size_t)nrows : cap;
    if (give) HIPCHK(c, hipMemcpyAsync(rows_out, h->d_out, sizeof(float) * give * N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller's buffer and rows_out are the caller's again
    if (nrows_out) *nrows_out = nrows;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

// A push of which nothing will ever be looked at: no vector it completes is a kept one (keep_one_in_n) and the partial vector it
// leaves behind is not one either.  Then nothing needs to reach the device - only the stream position moves on.  This is the
// common case of a sensor with a low sens_per_sec: spectrum_sensor_v2.py:86-87 keeps one vector in int(Sf / N / sens_per_sec)
// (97 of 98 at 1 MS/s, 1024 points, 10 PSDs per second), and GNU Radio's work() chunks hold 4-32 of them.
// -> true and the state advanced, or false and nothing touched.
static bool chain_push_dropped(oth_chain *h, size_t nsamples) {
    const size_t N = (size_t)h->nfft;
    size_t L = h->leftover, rest = nsamples;
    long long count = h->count;
    bool stale = h->leftover_stale;
    if (L) {
        const size_t take = rest < N - L ? rest : N - L;
        L += take;
        rest -= take;
        if (L == N) {
            if (--count == 0) {
                if (!stale) return false;      // the head vector is a kept one
                count = h->keep_n;
            }
            L = 0;
            stale = false;
        } else {
            if (count == 1 && !stale) return false;      // still inside a vector that will be kept: its samples are needed
            h->leftover = L;
            h->leftover_stale = true;      // (a dropped vector's samples: d_buf is not written)
            return true;
        }
    }
    const long long nvec = (long long)(rest / N);
    if (nvec > count - 1) return false;      // a vector of the body is kept
    count -= nvec;
    const size_t keep = rest - (size_t)nvec * N;
    if (keep && count == 1) return false;      // the vector that begins here will be kept
    h->leftover = keep;
    h->leftover_stale = keep != 0;
    h->count = (int)count;
    return true;
}

// sync_block.work() form: copy the scheduler's buffer into a pinned slot, enqueue H2D + kernels, record an event and return.
// The latest row is written by the last kernel straight into the slot's pinned host row (round 6: one stream operation less
// than a D2H copy behind it); the watcher collects it with oth_chain_poll / oth_chain_wait.
int oth_chain_push_async(oth_chain *h, const void *iq_host, size_t nsamples, uint64_t *ticket_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
    oth_ctx *c = h->ctx;
    if (!ticket_out) return fail(c, OTH_ERR_INVALID, "ticket_out is NULL");
    *ticket_out = 0;
    if (!iq_host && nsamples) return fail(c, OTH_ERR_INVALID, "iq is NULL");
    const int N = h->nfft;
    const uint64_t ticket = h->next_ticket;
    const int slot = (int)(ticket % oth_chain::kRing);
    h->ops = 0;
    if (nsamples && chain_push_dropped(h, nsamples)) {      // nothing to compute: no copy, no launch, no event
        h->noop[slot] = true;
        h->ticket_of[slot] = ticket;
        h->nrows_of[slot] = 0;
        h->next_ticket = ticket + 1;
        *ticket_out = ticket;
        return OTH_OK;
    }
    if (use_device(c)) return OTH_ERR_HIP;
    if (!h->ev[slot]) {
        HIPCHK(c, hipEventCreateWithFlags(&h->ev[slot], hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void **)&h->h_row[slot], sizeof(float) * N, hipHostMallocDefault));
    } else if (h->ticket_of[slot] && !h->noop[slot]) {
        HIPCHK(c, hipEventSynchronize(h->ev[slot]));      // only when the GPU is kRing pushes behind
    }
    h->noop[slot] = false;
    const size_t bytes = nsamples * sizeof(float2);
    const bool pinned_src = bytes > kPinnedStageMax && host_ptr_is_pinned(iq_host);
    const bool direct = bytes > kPinnedStageMax && !pinned_src;      // the runtime stages pageable memory itself
    const bool wait_copy = pinned_src && bytes > kPinnedRingMax;
    if (!direct && !wait_copy && h->h_in_cap[slot] < bytes) {
        if (h->h_in[slot]) HIPCHK(c, hipHostFree(h->h_in[slot]));
        h->h_in[slot] = nullptr;
        h->h_in_cap[slot] = 0;
        const size_t cap = bytes + bytes / 2 + 4096;
        HIPCHK(c, hipHostMalloc(&h->h_in[slot], cap, hipHostMallocDefault));
        h->h_in_cap[slot] = cap;
    }
    int rc;
    uint64_t nrows = 0;
    if (nsamples) {
        if ((rc = ensure(c, &h->d_stage, &h->stage_cap, bytes))) return rc;
        if (wait_copy) {
            if ((rc = copy_in_and_wait(c, h->d_stage, iq_host, bytes))) return rc;
        } else if (direct) {      // the runtime's staged copy returns once the caller's buffer has been read
            HIPCHK(c, hipMemcpyAsync(h->d_stage, iq_host, bytes, hipMemcpyHostToDevice, c->stream));
        } else {
            memcpy(h->h_in[slot], iq_host, bytes);      // the scheduler's buffer dies when work() returns
            HIPCHK(c, hipMemcpyAsync(h->d_stage, h->h_in[slot], bytes, hipMemcpyHostToDevice, c->stream));
        }
        h->ops += 1;
        // the latest row goes from the closing kernel straight into the slot's pinned row (device-visible host memory)
        if ((rc = chain_feed(h, h->d_stage, nsamples, h->h_row[slot], 1, &nrows))) return rc;
    }
    HIPCHK(c, hipEventRecord(h->ev[slot], c->stream));
    h->ticket_of[slot] = ticket;
    h->nrows_of[slot] = nrows;
    h->next_ticket = ticket + 1;
    *ticket_out = ticket;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

static int chain_collect(oth_chain *h, uint64_t ticket, float *row_out, uint64_t *nrows_out, int *ready, bool wait) {
    oth_ctx *c = h->ctx;
    const int slot = (int)(ticket % oth_chain::kRing);
    if (!ticket || h->ticket_of[slot] != ticket)
        return fail(c, OTH_ERR_STATE, "ticket unknown or overwritten (the ring keeps the last 4 pushes: latest wins)");
    if (h->noop[slot]) {      // the push enqueued nothing (all its vectors dropped): done when it returned
        if (ready) *ready = 1;
        if (nrows_out) *nrows_out = 0;
        return OTH_OK;
    }
    hipError_t e = wait ? hipEventSynchronize(h->ev[slot]) : hipEventQuery(h->ev[slot]);
    if (e == hipErrorNotReady) {
        if (ready) *ready = 0;
        return OTH_OK;
    }
    if (e != hipSuccess) return fail(c, OTH_ERR_HIP, std::string("event: ") + hipGetErrorString(e));
    if (ready) *ready = 1;
    if (nrows_out) *nrows_out = h->nrows_of[slot];
    if (row_out && h->nrows_of[slot]) memcpy(row_out, h->h_row[slot], sizeof(float) * h->nfft);
    return OTH_OK;
}

int oth_chain_poll(oth_chain *h, uint64_t ticket, float *row_out, uint64_t *nrows_out, int *ready) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h || !ready) return fail(h ? h->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    *ready = 0;
    return chain_collect(h, ticket, row_out, nrows_out, ready, false);
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_wait(oth_chain *h, uint64_t ticket, float *row_out, uint64_t *nrows_out) {
    OTH_TRY
    // the wait itself runs WITHOUT the context lock: work() on the scheduler thread must be able to enqueue meanwhile
    hipEvent_t ev = nullptr;
    {
        CtxGuard guard_(h ? h->ctx : nullptr);
        if (!h) return fail(nullptr, OTH_ERR_INVALID, "chain is NULL");
        const int slot = (int)(ticket % oth_chain::kRing);
        if (!ticket || h->ticket_of[slot] != ticket)
            return fail(h->ctx, OTH_ERR_STATE, "ticket unknown or overwritten (the ring keeps the last 4 pushes: latest wins)");
        if (h->noop[slot]) {
            if (nrows_out) *nrows_out = 0;
            return OTH_OK;
        }
        ev = h->ev[slot];
    }
    hipError_t e = hipEventSynchronize(ev);
    if (e != hipSuccess) return fail(h->ctx, OTH_ERR_HIP, std::string("event: ") + hipGetErrorString(e));
    CtxGuard guard_(h->ctx);
    int ready = 0;
    return chain_collect(h, ticket, row_out, nrows_out, &ready, false);
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_ticket_rows(oth_chain *h, uint64_t ticket, uint64_t *nrows_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h || !nrows_out) return fail(h ? h->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    const int slot = (int)(ticket % oth_chain::kRing);
    if (!ticket || h->ticket_of[slot] != ticket)
        return fail(h->ctx, OTH_ERR_STATE, "ticket unknown or overwritten (the ring keeps the last 4 pushes: latest wins)");
    *nrows_out = h->nrows_of[slot];
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_last_push_ops(oth_chain *h, uint64_t *ops_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h || !ops_out) return fail(h ? h->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    *ops_out = h->ops;
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_get_peak(oth_chain *h, float *peak_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h || !peak_out) return fail(h ? h->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    oth_ctx *c = h->ctx;
    HIPCHK(c, hipMemcpyAsync(peak_out, h->d_peak, sizeof(float) * h->nfft, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_chain_get_iir(oth_chain *h, float *lin_out) {
    OTH_TRY
    CtxGuard guard_(h ? h->ctx : nullptr);
    if (!h || !lin_out) return fail(h ? h->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    oth_ctx *c = h->ctx;
    HIPCHK(c, hipMemcpyAsync(lin_out, h->d_iir, sizeof(float) * h->nfft, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH((h ? h->ctx : nullptr))
}

int oth_rows_group_mean(oth_ctx *c, const float *rows_host, size_t nrows, int nfft, int group, float *out_host) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !rows_host || !out_host || nfft < 1 || group < 1 || nrows < (size_t)group)
        return fail(c, OTH_ERR_INVALID, "bad argument");
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t ngroups = nrows / group;
    const size_t in_bytes = sizeof(float) * ngroups * group * nfft, out_bytes = sizeof(float) * ngroups * nfft;
    int rc = ensure(c, &c->scratch, &c->scratch_cap, in_bytes + out_bytes);
    if (rc) return rc;
    float *d_in = (float *)c->scratch, *d_out = (float *)(c->scratch + in_bytes);
    HIPCHK(c, hipMemcpyAsync(d_in, rows_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_group_mean(d_in, (long long)ngroups, nfft, group, d_out, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_host, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

static size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

// -> device pointers to the channel slice bounds, uploading them only when they differ from the cached copy
static int channel_bounds_dev(oth_ctx *c, int nch, const int *lo, const int *hi, const int **dlo, const int **dhi) {
    *dlo = *dhi = nullptr;
    if (nch <= 0) return OTH_OK;
    const size_t n = 2 * (size_t)nch;
    bool same = c->d_bounds && c->bounds_host.size() == n;
    for (int i = 0; same && i < nch; ++i) same = c->bounds_host[i] == lo[i] && c->bounds_host[nch + i] == hi[i];
    if (!same) {
        if (n > c->bounds_cap) {
            if (c->bounds_ev) HIPCHK(c, hipEventSynchronize(c->bounds_ev));
            HIPCHK(c, hipStreamSynchronize(c->stream));      // kernels may still read the old device copy
            if (c->d_bounds) hipFree(c->d_bounds);
            if (c->h_bounds) hipHostFree(c->h_bounds);
            c->d_bounds = c->h_bounds = nullptr;
            c->bounds_cap = 0;
            c->bounds_host.clear();
            if (hipMalloc(&c->d_bounds, sizeof(int) * n) != hipSuccess) return fail(c, OTH_ERR_NOMEM, "device allocation failed");
            HIPCHK(c, hipHostMalloc((void **)&c->h_bounds, sizeof(int) * n, hipHostMallocDefault));
            c->bounds_cap = n;
        }
        if (!c->bounds_ev) HIPCHK(c, hipEventCreateWithFlags(&c->bounds_ev, hipEventDisableTiming));
        else HIPCHK(c, hipEventSynchronize(c->bounds_ev));      // the previous upload has read the pinned words
        memcpy(c->h_bounds, lo, sizeof(int) * nch);
        memcpy(c->h_bounds + nch, hi, sizeof(int) * nch);
        HIPCHK(c, hipMemcpyAsync(c->d_bounds, c->h_bounds, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->bounds_ev, c->stream));
        c->bounds_host.assign(c->h_bounds, c->h_bounds + n);
    }
    *dlo = c->d_bounds;
    *dhi = c->d_bounds + nch;
    return OTH_OK;
}

int oth_channel_power(oth_ctx *c, const float *psd_host, int nfft, double srch_bins, int nch, const int *lo,
                      const int *hi, float *power_out, float *movavg_out) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !psd_host || !lo || !hi || !power_out || nfft < 1 || nch < 1 || !(srch_bins >= 1.0))
        return fail(c, OTH_ERR_INVALID, "bad argument (srch_bins must be >= 1)");
    for (int i = 0; i < nch; ++i)
        if (lo[i] < 0 || hi[i] > nfft) return fail(c, OTH_ERR_INVALID, "channel slice outside [0, nfft]");
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t o_psd = 0, o_ma = up16(o_psd + sizeof(float) * nfft), o_maf = up16(o_ma + sizeof(double) * nfft),
                 o_lo = up16(o_maf + sizeof(float) * nfft), o_hi = up16(o_lo + sizeof(int) * nch),
                 o_pw = up16(o_hi + sizeof(int) * nch), bytes = up16(o_pw + sizeof(float) * nch);
    int rc = ensure(c, &c->scratch, &c->scratch_cap, bytes);
    if (rc) return rc;
    unsigned char *d = c->scratch;
    HIPCHK(c, hipMemcpyAsync(d + o_psd, psd_host, sizeof(float) * nfft, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + o_lo, lo, sizeof(int) * nch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + o_hi, hi, sizeof(int) * nch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_channel_power((const float *)(d + o_psd), 1, nfft, srch_bins, nch, (const int *)(d + o_lo),
                                   (const int *)(d + o_hi), (double *)(d + o_ma), (float *)(d + o_pw),
                                   (float *)(d + o_maf), c->stream));
    HIPCHK(c, hipMemcpyAsync(power_out, d + o_pw, sizeof(float) * nch, hipMemcpyDeviceToHost, c->stream));
    if (movavg_out)
        HIPCHK(c, hipMemcpyAsync(movavg_out, d + o_maf, sizeof(float) * nfft, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_bin_threshold(oth_ctx *c, const float *psd_host, int nrows, int nfft, double srch_bins, float thr_leveler,
                      unsigned char *mask_out, float *noise_out) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !psd_host || !mask_out || nrows < 1 || nfft < 1 || !(srch_bins >= 1.0))
        return fail(c, OTH_ERR_INVALID, "bad argument (srch_bins must be >= 1)");
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t nb = (size_t)nrows * nfft;
    const size_t o_mask = sizeof(float) * nb, o_noise = up16(o_mask + nb);
    int rc = ensure(c, &c->scratch, &c->scratch_cap, o_noise + sizeof(float) * nrows);
    if (rc) return rc;
    unsigned char *d = c->scratch;
    HIPCHK(c, hipMemcpyAsync(d, psd_host, sizeof(float) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_bin_threshold((const float *)d, nrows, nfft, srch_bins, thr_leveler, d + o_mask,
                                   (float *)(d + o_noise), c->stream));
    HIPCHK(c, hipMemcpyAsync(mask_out, d + o_mask, nb, hipMemcpyDeviceToHost, c->stream));
    if (noise_out)
        HIPCHK(c, hipMemcpyAsync(noise_out, d + o_noise, sizeof(float) * nrows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

// Decision stage of the batched scanner on PSD rows that are already in HBM (BASELINE config 5): one launch
// sequence, context-owned scratch, no copy of the rows.  Host results: mask (nullable), noise[nrows],
// power[nrows][nch] (nullable when nch == 0).
int oth_scan_decide_dev(oth_ctx *c, const float *psd_rows_dev, int nrows, int nfft, double srch_bins, float thr_leveler,
                        int nch, const int *lo, const int *hi, unsigned char *mask_out, float *noise_out,
                        float *power_out) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !psd_rows_dev || nrows < 1 || nfft < 1 || nch < 0 || !(srch_bins >= 1.0) || (nch && (!lo || !hi || !power_out)))
        return fail(c, OTH_ERR_INVALID, "bad argument (srch_bins must be >= 1)");
    for (int i = 0; i < nch; ++i)
        if (lo[i] < 0 || hi[i] > nfft) return fail(c, OTH_ERR_INVALID, "channel slice outside [0, nfft]");
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t nb = (size_t)nrows * nfft;
    const size_t o_ma = 0, o_mask = up16(o_ma + sizeof(double) * nb), o_noise = up16(o_mask + nb),
                 o_pw = up16(o_noise + sizeof(float) * nrows), o_tm = up16(o_pw + sizeof(float) * nrows * (nch + 1)),
                 bytes = up16(o_tm + sizeof(float) * nrows * scan_decide_tiles(nfft));
    int rc = ensure(c, &c->scratch, &c->scratch_cap, bytes);
    if (rc) return rc;
    unsigned char *d = c->scratch;
    const int *dlo = nullptr, *dhi = nullptr;
    if ((rc = channel_bounds_dev(c, nch, lo, hi, &dlo, &dhi))) return rc;
    HIPCHK(c, launch_scan_decide(psd_rows_dev, nrows, nfft, srch_bins, thr_leveler, nch, dlo,
                                 dhi, (double *)(d + o_ma), (float *)(d + o_tm), mask_out ? d + o_mask : nullptr,
                                 (float *)(d + o_noise), nch ? (float *)(d + o_pw) : nullptr, c->stream));
    if (mask_out) HIPCHK(c, hipMemcpyAsync(mask_out, d + o_mask, nb, hipMemcpyDeviceToHost, c->stream));
    if (noise_out)
        HIPCHK(c, hipMemcpyAsync(noise_out, d + o_noise, sizeof(float) * nrows, hipMemcpyDeviceToHost, c->stream));
    if (nch)
        HIPCHK(c, hipMemcpyAsync(power_out, d + o_pw, sizeof(float) * nrows * nch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

int oth_scan_decide_dev_out(oth_ctx *c, const float *psd_rows_dev, int nrows, int nfft, double srch_bins, float thr_leveler,
                            int nch, const int *lo, const int *hi, unsigned char *mask_dev, float *noise_dev,
                            float *power_dev) {
    OTH_TRY
    CtxGuard guard_(c);
    if (!c || !psd_rows_dev || !noise_dev || nrows < 1 || nfft < 1 || nch < 0 || !(srch_bins >= 1.0) ||
        (nch && (!lo || !hi || !power_dev)))
        return fail(c, OTH_ERR_INVALID, "bad argument (srch_bins must be >= 1)");
    for (int i = 0; i < nch; ++i)
        if (lo[i] < 0 || hi[i] > nfft) return fail(c, OTH_ERR_INVALID, "channel slice outside [0, nfft]");
    if (use_device(c)) return OTH_ERR_HIP;
    const size_t nb = (size_t)nrows * nfft;
    const size_t o_tm = up16(sizeof(double) * nb), bytes = up16(o_tm + sizeof(float) * nrows * scan_decide_tiles(nfft));
    int rc = ensure(c, &c->scratch, &c->scratch_cap, bytes);
    if (rc) return rc;
    unsigned char *d = c->scratch;
    const int *dlo = nullptr, *dhi = nullptr;      // cached on the device: no host copy on the steady-state path
    if ((rc = channel_bounds_dev(c, nch, lo, hi, &dlo, &dhi))) return rc;
    HIPCHK(c, launch_scan_decide(psd_rows_dev, nrows, nfft, srch_bins, thr_leveler, nch, dlo, dhi, (double *)d,
                                 (float *)(d + o_tm), mask_dev,
                                 noise_dev, nch ? power_dev : nullptr, c->stream));
    return OTH_OK;
    OTH_CATCH(c)
}

// xcorr / fac at the lengths the one-workgroup kernel does not take: np.fft calls composed from any_fft_nat()
static int xcorr_any(oth_ctx *c, const void *a, size_t na, const void *b, size_t nb, int L, float *out, int mode) {
    AnyTables t;
    int rc = any_tables_init(c, L, &t);
    if (rc) return rc;
    const size_t nsc = any_fft_nat_scratch(t.sh), nout = (size_t)(L - L / 2);
    rc = ensure(c, &c->scratch, &c->scratch_cap, sizeof(float2) * (2 * (size_t)L + nsc) + sizeof(float) * nout);
    if (!rc) {
        float2 *A = (float2 *)c->scratch, *Bv = A + L, *sc = Bv + L;
        float *o = (float *)(sc + nsc);
        auto run = [&]() -> int {
            int r;
            HIPCHK(c, hipMemsetAsync(A, 0, sizeof(float2) * 2 * (size_t)L, c->stream));
            HIPCHK(c, hipMemcpyAsync(A, a, sizeof(float2) * na, hipMemcpyHostToDevice, c->stream));
            if ((r = any_fft_nat(c, t, A, sc))) return r;                                                  // e = fft(a, L)
            if (mode == 0) {
                HIPCHK(c, hipMemcpyAsync(Bv, b, sizeof(float2) * nb, hipMemcpyHostToDevice, c->stream));
                if ((r = any_fft_nat(c, t, Bv, sc))) return r;                                             // f = fft(b, L)
                HIPCHK(c, launch_any_ew(4, A, A, Bv, nullptr, L, L, 0, 0, c->stream));                    // conj(f conj(e)) = conj(f) e
                if ((r = any_fft_nat(c, t, A, sc))) return r;                                              // = L conj(ifft(f conj(e)))
                HIPCHK(c, launch_any_abs(o, A, (int)nout, 1.0f / (float)L, c->stream));                   // |fftshift(h)[L/2:]| = |h[:L - L/2]|
            } else {
                HIPCHK(c, launch_any_ew(5, A, A, nullptr, nullptr, L, L, 0, 0, c->stream));               // |fft(d, L)|
                if ((r = any_fft_nat(c, t, A, sc))) return r;
                HIPCHK(c, launch_any_abs(o, A, (int)nout, 1.0f, c->stream));
            }
            HIPCHK(c, hipMemcpyAsync(out, o, sizeof(float) * nout, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return OTH_OK;
        };
        rc = run();
    }
    if (rc) hipStreamSynchronize(c->stream);
    any_tables_free(t);
    return rc;
}

static int xcorr_impl(oth_ctx *c, const void *a, size_t na, const void *b, size_t nb, int L, float *out, int mode) {
    if (!c || !a || !out || (mode == 0 && !b)) return fail(c, OTH_ERR_INVALID, "bad argument");
    if (L < 1) return fail(c, OTH_ERR_INVALID, "L must be positive");
    if (na > (size_t)L) na = (size_t)L;      // np.fft.fft(a, L) keeps the first L samples of a longer input
    if (nb > (size_t)L) nb = (size_t)L;
    if (use_device(c)) return OTH_ERR_HIP;
    if (!generic_supported(L)) return xcorr_any(c, a, na, b, nb, L, out, mode);
    const float2 *tw = nullptr;
    int rc = get_twiddles(c, L, &tw);
    if (rc) return rc;
    if ((rc = ensure(c, &c->scratch, &c->scratch_cap, sizeof(float2) * 3 * (size_t)L))) return rc;
    float2 *d = (float2 *)c->scratch;
    HIPCHK(c, hipMemsetAsync(d, 0, sizeof(float2) * 3 * L, c->stream));
    HIPCHK(c, hipMemcpyAsync(d, a, sizeof(float2) * na, hipMemcpyHostToDevice, c->stream));
    if (mode == 0) HIPCHK(c, hipMemcpyAsync(d + L, b, sizeof(float2) * nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_xcorr(L, d, d + L, tw, (float *)(d + 2 * L), mode, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, d + 2 * L, sizeof(float) * (L - L / 2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
}

int oth_xcorr(oth_ctx *c, const void *a, size_t na, const void *b, size_t nb, int L, float *out) {
    OTH_TRY
    CtxGuard guard_(c);
    return xcorr_impl(c, a, na, b, nb, L, out, 0);
    OTH_CATCH(c)
}

#ifdef OTH_EXPERIMENTS      // the three readers below serve the stamped / diagnostic kernel builds: `make EXP=1` only
// Not part of the ABI (not in the header): raw bytes behind the partial sums (diagnostic kernel builds).
int oth__debug_tail(oth_plan *p, void *out, size_t nbytes, int *nwg) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p || !out || !nwg) return OTH_ERR_INVALID;
    oth_ctx *c = p->ctx;
    if (nbytes > 1024 * (size_t)p->last_W) return OTH_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(out, p->d_partial + (size_t)p->last_W * p->nfft, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nwg = p->last_W;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

// Not part of the ABI (not in the header): reads the per-workgroup stamps of the diagnostic kernel build.
int oth__debug_stamps(oth_plan *p, unsigned long long *out, int max_wg, int *nwg) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p || !out || !nwg) return OTH_ERR_INVALID;
    oth_ctx *c = p->ctx;
    const int n = p->last_W < max_wg ? p->last_W : max_wg;
    if (n != p->last_W) return OTH_ERR_INVALID;      // records [n][4] then phases [n][4 waves][12]
    HIPCHK(c, hipMemcpyAsync(out, p->d_partial + (size_t)p->last_W * p->nfft, (32 + 384) * (size_t)n,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nwg = n;
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

// Not part of the ABI: raw bytes of the plan's partial-sum buffer from a float offset on (diagnostic builds' stamps).
int oth__debug_partial_raw(oth_plan *p, size_t float_offset, void *out, size_t nbytes) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p || !out) return OTH_ERR_INVALID;
    oth_ctx *c = p->ctx;
    if (float_offset * sizeof(float) + nbytes > p->partial_cap) return OTH_ERR_INVALID;
    HIPCHK(c, hipMemcpyAsync(out, p->d_partial + float_offset, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

#endif      // OTH_EXPERIMENTS

// The launch recipe of a plan shape as text, WITHOUT a device (pure host logic; runtime_occupancy = 0 takes the resident
// workgroups per CU from the built-in MI355X table, 1 asks the occupancy calculator and needs a GPU).
//   window_class: 0 all ones, 1 spectrum confined (periodic cosine-sum windows: both detrend tables exist), 2 wide
//   (e.g. a symmetric Hamming: no table), 3 confined to 256 F bins but not to |k| < 16 (16384 points only)
//   detrend_mode: OTH_DETREND_*;  kernel_pref: OTH_KERNEL_*;  sched_pref: OTH_SCHED_*;  variant: as oth_plan_set_tuning
int oth__debug_recipe(int nfft, int nperseg, int noverlap, int window_class, int detrend_mode, int two_channel, int kernel_pref,
                      const char *variant, int sched_pref, long long nseg, int nstreams, int cu_count, int runtime_occupancy,
                      char *buf, size_t buflen) {
    OTH_TRY
    if (!buf || !buflen || nperseg < 1 || nperseg > nfft || noverlap < 0 || noverlap >= nperseg || nseg < 1 || nstreams < 1)
        return fail(nullptr, OTH_ERR_INVALID, "bad argument");
    PlanShape sh;
    sh.nfft = nfft;
    sh.nperseg = nperseg;
    sh.step = nperseg - noverlap;
    sh.detrend = detrend_mode != OTH_DETREND_NONE;
    sh.fast_detrend = detrend_mode == OTH_DETREND_CONSTANT_FAST;
    // the tables oth_welch_plan builds: 4096 / 2048 / 8192 / 16384 points, nperseg = nfft, a confined window spectrum
    const bool table_size = (nfft == 4096 || nfft == 2048 || nfft == 8192 || nfft == 16384) && nperseg == nfft;
    sh.fd_ok = sh.detrend && table_size && (window_class == 0 || window_class == 1 || window_class == 3);
    sh.fd1x_ok = sh.detrend && (nfft == 16384 || nfft == 8192) && nperseg == nfft && (window_class == 0 || window_class == 1);
    sh.rect_window = window_class == 0;
    sh.kernel = kernel_pref;
    sh.sched = sched_pref;
    sh.tune_variant = variant ? variant : "";
    if (!generic_supported(nfft) && any_describe(nfft, &sh.any))
        return fail(nullptr, OTH_ERR_UNSUPPORTED, "transform length outside [1, 1048576] (Bluestein: n <= 524288)");
    LaunchRecipe r;
    const char *why = "";
    if (int rc = resolve_recipe(sh, two_channel != 0, nseg, nstreams, cu_count > 0 ? cu_count : 256,
                                runtime_occupancy ? runtime_bpc : table_bpc, &r, &why))
        return fail(nullptr, rc, why);
    snprintf(buf, buflen, "%s", recipe_text(r, nfft).c_str());
    return OTH_OK;
    OTH_CATCH(nullptr)
}

// recipe of the plan's last averaging launch ("" before the first)
int oth__debug_last_recipe(oth_plan *p, char *buf, size_t buflen) {
    OTH_TRY
    CtxGuard guard_(p ? p->ctx : nullptr);
    if (!p || !buf || !buflen) return fail(p ? p->ctx : nullptr, OTH_ERR_INVALID, "bad argument");
    snprintf(buf, buflen, "%s", p->last_recipe.c_str());
    return OTH_OK;
    OTH_CATCH((p ? p->ctx : nullptr))
}

int oth_fac(oth_ctx *c, const void *data, size_t n, int L, float *out) {
    OTH_TRY
    CtxGuard guard_(c);
    return xcorr_impl(c, data, n, nullptr, 0, L, out, 1);
    OTH_CATCH(c)
}

}  // extern "C"
