// Internal declarations shared by the HIP translation units of libofdmtools_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace oth {

// Launch description of the segment-averaging (Welch / CSD) kernels.
constexpr int kPilotProbes = 8;      // segment means per stream behind WelchArgs.pilot / SegArgs.pilot

struct WelchArgs {
    const float2 *x;        // device IQ, stream 0
    const float2 *y;        // second channel (CSD) or nullptr
    const float *win;       // device window, nperseg floats
    const float2 *tw;       // device twiddle table W_nfft^k, nfft entries
    float *partial;         // [nstreams][wg_per_stream][nch][nfft] partial sums
    long long nseg;         // segments per stream
    size_t stream_stride;   // samples between streams
    int nperseg;
    int step;               // nperseg - noverlap
    int detrend;
    int wg_per_stream;
    int nstreams;
    // welch4096 segment schedule: 0 contiguous, 1 interleaved chunks, 2 dynamic chunk queue
    int sched;
    int chunk;              // segments per chunk (sched 1, 2)
    int tail_chunk;         // segments per chunk after the first nbig chunks
    long long nbig;         // number of full-size chunks
    unsigned *queue;        // [nstreams] tickets, zeroed before the launch (sched 2)
    // welch4096ws frequency-domain detrend: per consumer thread t = 16 k0 + k1 the window spectrum at bins
    // k0 + 16 k1 and k0 + 16 k1 + 3840 (re, im, re, im); nullptr when the window's spectrum is not confined
    const float4 *fd;
    // Pilot of the constant detrend (round 4; every plan but OTH_DETREND_CONSTANT_FAST): per stream (x streams, then y
    // streams) kPilotProbes segment means spread over the launch (launch_pilot_mean); load_pilot() averages them into one
    // complex value NEAR the stream's mean.  The PILOT build of every detrending kernel subtracts it from each sample
    // as it is loaded, so the sums, the window products and (in the builds that detrend after the transform) the
    // transform itself see x - pilot: no DC line whose float32 rounding would stay behind, and the segment mean that is
    // removed afterwards is a small residual known to its own rounding.  The result is mathematically unchanged (a
    // constant detrend removes any constant); numerically bins 0, +-1 go from the float32 mean's 1e-4 ... 5e-3 (SciPy
    // on complex64 included) to 1e-6.  nullptr: the builds without.
    const float2 *pilot;
    // 1 (round 5; the role-split 4096-point kernels, one- and two-channel): the PILOT build forms its pilot itself in
    // the launch's prologue - every producer team adds eight probes of 256 consecutive samples spread over its stream
    // (inline_pilot() in fft4096.hip.h) - instead of reading WelchArgs.pilot: no pilot_mean_kernel launch in front of the
    // transform (5.9 us + a dependent-launch gap per step).  The pilot only has to be NEAR the mean; every workgroup reads
    // the same samples in the same order, so all hold the same bits.
    int pilot_inline;
};

// Launch description of segfft.hip: segment transforms of 1024 / 2048 / 4096 points by teams of nfft / 16
// threads, Welch average or periodogram chain (see the file header).
struct SegArgs {
    const float2 *x;        // device IQ, stream 0
    size_t stream_stride;   // samples between streams
    int nstreams;
    const float *win;       // nfft floats
    const float2 *tw;       // W_nfft^k, nfft entries
    const float4 *fd;       // role-split build, frequency-domain detrend: [nfft / 16 threads][16 / R] window-spectrum pairs
    const float2 *pilot;    // per stream, as WelchArgs.pilot (nullptr: none)
    long long first;        // sample index of segment 0 (chain: first kept vector)
    long long step;         // samples between segment starts (chain: keep_n * nfft)
    long long nseg;         // segments per stream
    int detrend;
    int chain;              // 0: Welch average (sum |X|^2 of every segment); 1: periodogram chain
    float *partial;         // [nstreams][wg_per_stream][nfft] accumulator rows, natural bin order (or nullptr)
    // chain only
    int acc_mode;           // 1 weighted sum (IIR), 2 max (peak hold), 3 none
    long long acc_end;      // segments s < acc_end take part in the accumulation
    float l2;               // log2(1 - alpha): weight of segment s is 2^(l2 (acc_end - 1 - s))
    float *rows;            // [nstreams][nseg - store_from][nfft]: epilogue values of segments s >= store_from
    long long store_from;
    int epilogue;           // OTH_EPI_*
    float scale;            // applied to |X|^2 (MAG2_OVER_N2)
    int fftshift;
    // schedule (as WelchArgs)
    int wg_per_stream;
    int sched;
    int chunk;
    int tail_chunk;
    long long nbig;
    unsigned *queue;
};

struct PgramArgs {
    const float2 *x;
    const float *win;       // nfft floats
    const float2 *tw;
    float *rows;            // [nrows][nfft]
    long long first_vec;    // index of the first kept vector in x
    long long nrows;
    int keep_n;
    int fftshift;
    int epilogue;
    float scale;            // applied to |X|^2 (OVER_N2)
};

constexpr int kReduceGroups = 16;   // row groups of the two-stage partial-sum reduction
// Many SHORT partial rows (256 ... 1024 points: 2048 ... 12288 teams' rows of 1 ... 4 KiB): the one-launch reduction has
// nfft / 16 = 16 ... 64 workgroups to read 8-12 MiB with (63 / 33 / 13 us at 256 / 512 / 1024 points, late round 5).
// There the rows go through `finalize_row_groups(nfft, W)` groups first - enough for 256 workgroups - and the one-launch
// kernel then takes the group rows.  0: not this shape.
inline int finalize_row_groups(int nfft, int W, int nch) {
    if (nch != 1 || nfft > 1024 || (nfft % 256) != 0 || W < 1024) return 0;
    return 65536 / nfft;      // (nfft / 256) x groups = 256 workgroups: 256 / 128 / 64 groups
}

struct FinalizeArgs {
    const float *partial;   // [nstreams][W][nch][nfft]
    float *scratch;         // [nstreams][kReduceGroups][nch][nfft] or nullptr (single-stage)
    float *out0;            // psd / pxx      [nstreams][nout]
    float *out1;            // pyy  (CSD)
    float *out2;            // pxy  interleaved (CSD)
    float *out3;            // cxy  (CSD)
    double scale;
    int W;
    int nfft;
    int nch;                // 1 or 4
    int layout;             // 0 natural, 1 welch4096 digit order, 2 / 3 welch16k order at 16384 / 8192, 4 / 5 welch16k1x order at 16384 / 8192,
                            // 6 the two-level any-length route: position k1 l2 + k2 holds bin k1 + l1 k2 (kernels_misc.hip bin_pos)
    int l1, l2;             // layout 6 only
    int fftshift;
    int trim;
    int db;
    int nout;
    int accumulate;         // out0 += (raw sums; streaming form)
    unsigned *queue_reset;  // chunk-ticket counters of the averaging kernel, zeroed here for the next launch (or null)
    int queue_n;
    // Completion word for a host that polls instead of sleeping in hipStreamSynchronize (oth_welch_exec / _poll / _wait,
    // round 5): the outputs go straight to pinned host memory; every block makes its stores visible at system scope and
    // arrives on done_count (device; the last block re-arms it to 0), and the last one to arrive then stores seq_value
    // into *host_seq (pinned).  nullptr: none of this runs.
    unsigned *done_count;
    unsigned *host_seq;
    unsigned seq_value;
};

// Every launcher returns hipSuccess / error of the launch only (asynchronous).
hipError_t launch_welch_generic(int nfft, const WelchArgs &a, hipStream_t s);
// welch4096.hip is built once per variant tag (Makefile W4096_VARIANTS); nfft 4096, nperseg 4096, y == nullptr
#define OTH_DECL_W4096(tag)                                                     \
    hipError_t launch_welch_tuned4096_##tag(const WelchArgs &a, hipStream_t s); \
    int tuned4096_blocks_per_cu_##tag();
OTH_DECL_W4096(dpp)
#ifdef OTH_EXPERIMENTS
OTH_DECL_W4096(diag)
OTH_DECL_W4096(exp1)
OTH_DECL_W4096(exp2)
OTH_DECL_W4096(exp3)
OTH_DECL_W4096(exp4)
#endif
OTH_DECL_W4096(pipe)
// welch4096ws.hip: wave-specialised producer/consumer form; step 2048, detrend needs WelchArgs.fd
// (built once per tag like welch4096.hip)
OTH_DECL_W4096(ws)
// welch4096ws2.hip: the same arithmetic in one 1024-thread workgroup per CU, two runs of segments (A/B variant "ws2");
// WelchArgs.y = first sample of the second run, WelchArgs.nseg = segments per run, two partial rows per workgroup
OTH_DECL_W4096(ws2)
#ifdef OTH_EXPERIMENTS
OTH_DECL_W4096(wsx1)
OTH_DECL_W4096(wsx2)
OTH_DECL_W4096(wsx3)
OTH_DECL_W4096(wsx4)
#endif
// csd4096.hip: two-channel cross spectrum, nfft = nperseg = 4096
hipError_t launch_csd_tuned4096(const WelchArgs &a, hipStream_t s);
int csd4096_blocks_per_cu();
// csd4096ws.hip: the same as two wave-specialised pairs in one 1024-thread workgroup; step 2048, WelchArgs.fd for detrend
hipError_t launch_csd_tuned4096ws(const WelchArgs &a, hipStream_t s);
int csd4096ws_blocks_per_cu();
// welch16k.hip: nfft = nperseg = 16384 (one 1024-thread workgroup per CU) or 8192 (two 512-thread workgroups)
hipError_t launch_welch_tuned16k(int nfft, const WelchArgs &a, hipStream_t s);
// welch16k1x.hip: nfft = nperseg = 16384, no detrend, whole-segment loads (the scanner's non-overlapping vectors): one
// cross-wave exchange, two workgroup barriers per segment; partial rows in finalize layout 4
hipError_t launch_welch_tuned16k1x(int nfft, const WelchArgs &a, bool window, bool plain, hipStream_t s);
// the same transform at step = 8192 (50 % overlap, the kept half in registers); WelchArgs.fd = window_spectrum_table_16k1x
hipError_t launch_welch_tuned16k1x_half(int nfft, const WelchArgs &a, hipStream_t s);
hipError_t launch_welch_tuned8kws(const WelchArgs &a, hipStream_t s);      // 8192 points, 50 % overlap, role-split (contiguous runs only)
// the fused periodogram chain at 8192 / 16384 points (one workgroup per segment; workgroups per CU: 2 / 1)
hipError_t launch_chain16k(int nfft, const SegArgs &a, bool rect, hipStream_t s);
// the same chain at 16384 points on the one-exchange pipelined loop (welch16k1x.hip); partial rows in layout 4; needs
// chunks of at least two segments
hipError_t launch_chain16k1x(int nfft, const SegArgs &a, bool rect, hipStream_t s);
hipError_t launch_pgram(int nfft, const PgramArgs &a, hipStream_t s);
// segfft.hip
bool seg_supported(int nfft);
// kind: 0 Welch with step = nfft / 2 (the overlapped half stays in registers), 1 Welch with any step, 2 chain
int seg_teams_per_cu(int nfft, int kind, bool wps4);
hipError_t launch_seg(int nfft, const SegArgs &a, int kind, bool wps4, hipStream_t s);
// zero-padded segments (nperseg = nfft / 4 or nfft / 2 at nfft = 1024 / 2048: the sweeper's call at those sizes)
bool seg_padded_supported(int nfft, int nperseg);
int seg_padded_teams_per_cu(int nfft, int nperseg, int kind);
hipError_t launch_seg_padded(int nfft, int nperseg, const SegArgs &a, int kind, hipStream_t s);
int segws_teams_per_cu(int nfft);
hipError_t launch_segws(int nfft, const SegArgs &a, int det, hipStream_t s);
// partial rows of a chain launch + the stored raw rows -> IIR / peak state and the rows handed back; `scratch` holds
// chain_tail_groups(W, nfft) rows of nfft floats (0 rows: not needed)
int chain_tail_groups(int W, int nfft);
// layout: bin order of the partial rows (kernels_misc.hip bin_pos: 0 natural, 2 / 3 welch16k order at 16384 / 8192)
hipError_t launch_chain_tail(const float *partial, float *scratch, int W, int nfft, int layout, int fftshift, int acc_mode, long long nbase,
                             float alpha, float kdb, float *iir_state, float *peak_state, const float *raw_rows,
                             long long nraw, float *rows_out, hipStream_t s);
hipError_t launch_set_flag(int *flag, int v, hipStream_t s);
hipError_t launch_finalize(const FinalizeArgs &a, int nstreams, hipStream_t s);
hipError_t launch_scale(const float *sum, float *out, int nfft, double scale, int fftshift, int trim, int db,
                        hipStream_t s);
hipError_t launch_csd_scale(const float *sums, int nfft, double scale, int fftshift, int trim, float *pxx, float *pyy,
                            float *pxy, float *cxy, hipStream_t s);
hipError_t launch_rows_epilogue(float *rows, long long nrows, int nfft, float alpha, float kdb, float *iir_state,
                                float *peak_state, int *peak_init, int do_iir, int do_peak, hipStream_t s);
hipError_t launch_group_mean(const float *rows, long long ngroups, int nfft, int group, float *out, hipStream_t s);
hipError_t launch_channel_power(const float *psd, int nrows, int nfft, double srch_bins, int nch, const int *lo,
                                const int *hi, double *movavg, float *power, float *movavg_f, hipStream_t s);
// tile_min: nrows x scan_decide_tiles(nfft) floats of scratch (the per-tile minima of the moving average)
int scan_decide_tiles(int nfft);
hipError_t launch_scan_decide(const float *psd, int nrows, int nfft, double srch_bins, float thr, int nch, const int *lo,
                              const int *hi, double *movavg, float *tile_min, unsigned char *mask, float *noise, float *power,
                              hipStream_t s);
hipError_t launch_bin_threshold(const float *psd, int nrows, int nfft, double srch_bins, float thr,
                                unsigned char *mask, float *noise, hipStream_t s);
hipError_t launch_xcorr(int L, const float2 *a, const float2 *b, const float2 *tw, float *out, int mode,
                        hipStream_t s);
hipError_t launch_synth(float2 *iq, size_t n, uint64_t seed, int ntones, const float *amp, const float *freq,
                        float dc_re, float dc_im, hipStream_t s);
hipError_t launch_read_probe(const void *p, size_t bytes, float *sink, hipStream_t s);
hipError_t launch_read_probe8(const void *p, size_t bytes, float *sink, hipStream_t s);
hipError_t launch_iq_power(const float2 *iq, size_t n, double *acc4, hipStream_t s);
// out[(channel * nstreams + stream) * kPilotProbes + k] = mean of the n samples of segment (nseg - 1) k / (kPilotProbes - 1)
// of stream `stream` of x (channel 0) and, when y != nullptr, of y (channel 1)
hipError_t launch_pilot_mean(const float2 *x, const float2 *y, size_t stream_stride, int n, long long step, long long nseg,
                             int nstreams, float2 *out, hipStream_t s);

bool generic_supported(int nfft);
size_t generic_lds_bytes(int nfft);
int generic_threads_for(int nfft);

// ---- fft_any.hip: transforms of any length (round 6) -------------------------------------------------------------------
constexpr int kAnyMaxPasses = 12;            // 2 * 3^8 = 13122 takes nine passes
constexpr int kAnyMaxTile = 16384;           // points of one LDS tile (128 KiB)
constexpr int kAnyMaxFft = 1 << 20;          // longest transform (and longest Bluestein M): L1 = 4096 rows of L2 = 256
enum AnyKind { ANY_NONE = 0, ANY_DIRECT, ANY_TWOLEVEL, ANY_BLUESTEIN, ANY_BLUESTEIN2 };

struct AnyShape {        // pure host description of a length (any_describe): needs no device
    int nfft;            // the transform the caller asked for
    int kind;            // AnyKind
    int L;               // length of the transforms that run (nfft, or Bluestein's M)
    int L1, L2, C;       // two-level split L = L1 L2 (L2 = row length) and columns per K1 / K3 tile
};
int any_describe(int nfft, AnyShape *out);      // 0, or -1 when the length is outside [1, kAnyMaxFft] (Bluestein: M too long)
bool any_smooth(int n);
int any_threads_for(int points);

struct AnyPass {
    int R;               // radix: 16, 8, 4, 2, 3, 5, 7
    int NS;              // length of the sub-transforms this pass combines
    int nbf;             // butterflies per column = n / R
    int inner;           // twiddle index step in units of W_n: n / (NS R)
    float inv_ns;
};
struct AnyFftDesc {
    int n, logC, npass;
    int tws;             // table order / n: W_n^j = tw[j tws]
    int tw_lds;          // 1: the kernel stages the n twiddles W_n^j in LDS behind the tile
    const float2 *tw;    // W_order^k
    AnyPass pass[kAnyMaxPasses];
};
void any_make_desc(int n, int C, const float2 *tw, int order, AnyFftDesc *d);

struct AnyArgs {
    AnyFftDesc f;
    // element (i, c) of tile t (blockIdx.x) of segment s sits at  i * es + t * tile_stride + c * cs  of the segment:
    // column tiles (cs = 1: C adjacent columns, es = the row length) or row tiles (es = 1, cs = the row length)
    int es, tile_stride, cs;
    float inv_n;               // 1 / f.n (row tiles)
    long long nseg;            // segments of this launch (rows blockIdx.y, blockIdx.y + gridDim.y, ...)
    // load 1 (stage): samples x[first + s seg_step + n], n < nperseg, (x - mean) * win [* chirp], zero padded
    int load_op;
    const float2 *x, *y;
    long long first, seg_step;
    int nperseg;
    const float *win;
    const float4 *mean;        // [channel][segment] (hi.re, hi.im, lo.re, lo.im) or nullptr
    size_t mean_ch_stride;
    const float2 *chirp;       // Bluestein c[n] or nullptr
    // load 0 / store 0: the workspace, [channel][segment][L]
    float2 *ws;
    size_t ws_seg_stride, ws_ch_stride;
    // mid 1: v = conj(v * midtab[nat]) and the transform again
    int mid_op;
    const float2 *midtab;
    // natural index of element (i, c) of tile t: i nat_i + t nat_t + c nat_c; position in a partial row likewise (pp_*)
    int nat_i, nat_t, nat_c, pp_i, pp_t, pp_c;
    // store 0: optional four-step twiddle twbig[i (t tw_t + c tw_c)]
    const float2 *twbig;
    int tw_t, tw_c;
    // store 1 / 2: partial[gridDim.y rows][channels][nbins]
    float *partial;
    int nbins;                 // bins kept (Bluestein: the first nfft of M)
    int first_chunk;           // overwrite (1) or add to (0) the partial rows
    int conj_out;              // the transform leaves conj(X) (Bluestein): matters for the cross term only
    // store 3: rows[s][nbins]
    float *rows;
    int epilogue, fftshift;
    float scale;
};
hipError_t launch_any_fft(const AnyArgs &a, int tiles, int gy, int gz, int store, hipStream_t s);
hipError_t launch_any_mean(const float2 *x, const float2 *y, long long first, long long seg_step, int nperseg, long long nseg,
                           float4 *out, size_t ch_stride, hipStream_t s);
hipError_t launch_any_ew(int op, float2 *dst, const float2 *src, const float2 *src2, const float2 *tab, int n, int nsrc, int L1,
                         int L2, hipStream_t s);
hipError_t launch_any_abs(float *out, const float2 *src, int n, float scale, hipStream_t s);

// ---- fft_tl.hip: the four-step route at 32768 / 65536 points on register radix-16 butterflies ---------------------------
struct TlArgs {
    const float2 *x;           // samples; segment s starts at x[first + s seg_step]
    const float2 *y;           // second channel (two-channel sums) or nullptr
    size_t aux_ch_stride;      // entries between the channels' means / sub-block sums
    size_t ws_ch_stride;       // points between the channels' workspaces
    long long first, seg_step;
    int nperseg;               // samples per segment (zero padded to L)
    const float *win;          // L window values, zero behind nperseg (oth_welch_plan's zero-extended table)
    const float4 *mean;        // per segment (hi.re, hi.im, lo.re, lo.im) or nullptr
    const double2 *bsum;       // or: sums of sub-blocks of kTlSub samples from x[first] on; segment s takes nsub of them
    int nsub, sub_step;        // from index s * sub_step
    float2 *ws;                // workspace [segment][k1][n2]
    size_t ws_seg_stride;
    long long nseg;
    const float2 *tw;          // W_L^k
    float *partial;            // [W][channels 1 | 4][L] sums, position k1 L2 + k2 (finalize layout 6)
    int first_chunk;
};
constexpr int kTlSub = 4096;
bool tl_supported(int L);
// welch32k.hip: 32768-point segments inside one workgroup
struct W32kArgs {
    const float2 *x;           // samples; segment s starts at x[first + s step]
    long long first, step;
    long long nseg;
    const float *win;          // 32768 window values
    const float2 *tw;          // W_32768^k, k < 32768
    float *partial;            // [W][32768] sums, finalize layout 7 (front: [W][65536], layout 8)
    int detrend;
    int front;                 // 1: 65536-point segments, a pair of workgroups each (win: 65536 values, tw: W_65536^k)
    const float *wpm;          // front + detrend: w[n] + w[n + 32768] (n < 32768), then w[n] - w[n + 32768]
};
int welch32k_rows(long long nseg, int cus, bool front = false);
hipError_t launch_welch32k(const W32kArgs &a, int W, hipStream_t s);
hipError_t launch_tl_blocksum(const float2 *x, long long first, long long nblocks, double2 *out, hipStream_t s);
hipError_t launch_tl_mean(const float2 *x, long long first, long long seg_step, int nperseg, long long nseg, float4 *out, hipStream_t s);
hipError_t launch_tl_k1(int L, const TlArgs &a, hipStream_t s);
hipError_t launch_tl_k2(int L, const TlArgs &a, int W, hipStream_t s);

}  // namespace oth
