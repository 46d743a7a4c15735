/*
 * ofdm_tools_hip.h - C ABI of libofdmtools_hip.so, the MI355X (gfx950) back end
 * of the gr-ofdm_tools spectrum-sensing hot path.
 *
 * The reference (gercap/gr-ofdm_tools) has NO native boundary of its own
 * (swig/ofdm_tools_swig.i:1-11 wraps nothing, python/__init__.py:45 leaves the
 * swig import commented out): its PSD arithmetic is delegated to GNU Radio C++
 * blocks and to scipy.signal.welch / numpy.fft from Python.  Each entry point
 * below therefore names the reference call site (file:line, relative to the
 * upstream tree) whose arithmetic it replaces.  The reference-side binding is a
 * ctypes stub; INTEGRATION.md shows it.
 *
 * Conventions
 *   - C linkage, plain C types, no C++/torch types in any signature.
 *   - Every function returns OTH_OK (0) or a negative OTH_ERR_* code; nothing
 *     throws or aborts: each entry point is a try/catch barrier, a C++
 *     exception raised below it (std::bad_alloc, std::system_error, ...) comes
 *     back as OTH_ERR_NOMEM / OTH_ERR_INTERNAL.  oth_last_error() gives the
 *     text for the last failure on that context (for a NULL context: the
 *     calling thread's last context-less failure).
 *   - IQ data is interleaved float32 (re, im) = numpy.complex64 = gr_complex.
 *   - The caller owns every buffer it passes.  The library owns contexts, plans
 *     and their device scratch.  A context wraps one device + one HIP stream.
 *     Entry points that take a context, or a plan / chain made from it, hold
 *     that context's lock for the duration of the call, so the blocks of one
 *     flowgraph (one scheduler thread each) may share a context; calls are then
 *     serialised and their kernels run in call order on its stream.
 *     oth_last_error() reports the context's most recent failure, whichever
 *     thread caused it.  oth_ctx_destroy() must not race with other calls.
 *   - "_dev" entry points take device pointers, are asynchronous on the
 *     context's stream and never synchronise; the others take host pointers
 *     (or a device source when src_is_device != 0), and return with the host
 *     output written.
 *   - No CPU fallback exists: without a usable GPU every compute entry point
 *     returns OTH_ERR_HIP.
 */
#ifndef OFDM_TOOLS_HIP_H
#define OFDM_TOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTH_ABI_VERSION 6      /* 3 = 2 + oth_chain_ticket_rows, oth_scan_decide_dev_out; 4 = 3 + OTH_ERR_INTERNAL,
                                  OTH_DETREND_CONSTANT_EXACT, OTH_DETREND_CONSTANT_FAST; 5 = 4 + oth_welch_exec_async /
                                  _poll / _wait; 6 = 5 + transform lengths outside the powers of two 64 ... 16384
                                  (TRANSFORM LENGTHS below), oth_plan_set_hostwait (additions only) */

/* TRANSFORM LENGTHS (ABI 6).  The reference puts no limit on a transform length - fft.fft_vcc(self.fft_len, ...)
 * (psd_logger.py:48, spectrum_sensor_v2.py:90, local_worker.py:62-63), sg.welch(nperseg=nFFT, nfft=nFFT)
 * (ofdm_cr_tools.py:214,322,342), np.fft.fft(., nFFT) (ofdm_cr_tools.py:177,157-160), fast_spectrum_scan's own
 * nFFT = 2^ceil(log2(npts)) (ofdm_cr_tools.py:474-475), the web gateway's free --nfft
 * (sdr_webserver/local_hw_gateway.py:284-285) - and neither do oth_welch_plan, oth_chain_create, oth_xcorr, oth_fac:
 *   powers of two 64 ... 16384             the tuned kernels and the radix-4 coverage kernels (as before)
 *   2-3-5-7-smooth lengths up to 16384     one launch, mixed-radix Stockham in LDS               "anyfft:direct"
 *   powers of two 32768 ... 1048576        four-step L1 x L2 through a workspace in HBM / L2      "anyfft:twolevel"
 *                                          (32768, 65536: register radix-16 kernels,              "anyfft:twolevel:r16")
 *   32768 / 65536, one channel,            the segment inside one workgroup's registers (65536:   "anyfft:onewg"
 *   nperseg = nfft                         a pair of workgroups, even / odd bins)
 *   every other length n <= 524288         Bluestein through 2^ceil(log2(2 n - 1)) points         "anyfft:bluestein[2]"
 * (the quoted names are what oth__debug_last_recipe reports).  Still refused, OTH_ERR_UNSUPPORTED with the reason in
 * oth_last_error(): powers of two above 1048576 and other lengths above 524288; OTH_KERNEL_TUNED on any of the new
 * lengths.  Semantics, tolerances and every other argument are unchanged; a constant detrend at these lengths is taken
 * from each segment's own mean (added in double, removed as a float pair) in every detrend mode. */

#define OTH_OK               0
#define OTH_ERR_INVALID     -1   /* bad argument */
#define OTH_ERR_HIP         -2   /* HIP runtime failure / no device */
#define OTH_ERR_UNSUPPORTED -3   /* size or mode not built */
#define OTH_ERR_NOMEM       -4
#define OTH_ERR_STATE       -5   /* call order (e.g. finalize with no data) */
#define OTH_ERR_INTERNAL    -6   /* a C++ exception was caught at the ABI (text in oth_last_error) */

/* detrend (scipy.signal.welch detrend=...) */
#define OTH_DETREND_NONE     0
#define OTH_DETREND_CONSTANT 1   /* per-segment mean removal, SciPy default - computed on x - pilot: before anything else
                                   every kernel takes a pilot value per stream (the average of eight probe means
                                   spread over the launch: formed in the prologue of the role-split 4096-point kernels,
                                   by one small launch in front of the others) off each sample as it is loaded,
                                   so no float32 arithmetic ever handles the DC line - neither the transform (the fastest
                                   2048 / 4096 / 8192 / 16384-point builds at 50 % overlap remove the mean after it,
                                   FFT(x w) - m FFT(w)) nor the segment mean, which becomes a small residual.
                                   Mathematically the same result; measured against float64: every
                                   bin inside 1e-4 at 1 ... 9 segments and |m| = 35 sigma, every bin at 2e-7 with 2047
                                   segments and a DC line of 3000 sigma (70 dB above the signal), bins 0, +-1 at 1e-6
                                   where SciPy on complex64 input reads 1e-4 ... 5e-3.  That is for a CONSTANT
                                   offset.  The pilot is one value per stream and launch - per workgroup where a
                                   workgroup walks one contiguous run of segments (launches of fewer than 32 segments
                                   per resident workgroup, OTH_SCHED_CONTIGUOUS: the 4096-point role-split kernel spreads
                                   its probes over its own run, round 6) - so an offset that MOVES by D within the reach
                                   of one pilot leaves a line of about D / 2 (a step: of D, in the one or two segments
                                   that hold it) to the float32 transform; the exact time-domain form sees the same
                                   line as real signal.  Measured over eight noise seeds at 2047 segments of 4096 points:
                                   every bin inside 1e-4 for a 3000-sigma opening transient (worst 8.9e-5; SciPy on
                                   complex64: 1.08e-4) and 2e-7 for drifts of up to 1200 sigma on the default plan; the
                                   time-domain builds within 4 ulp of the spectrum's peak amplitude on every bin and
                                   1e-4 on every bin at or above the median (tests/test_hip_parity.py
                                   test_pilot_under_a_transient_and_a_drifting_offset, profiles/r06_moving_offset.txt).
                                   Launches of fewer than 8 segments per stream detrend before the window (their own
                                   mean per segment); with 1-3 segments and an offset moving by 100 sigma every bin
                                   stays within 4 ulp of the row's peak amplitude (what a single float32 transform of a
                                   strong ramp leaves: 1e-4 ... 3e-4 of the weakest bins; SciPy on complex64: up to
                                   7e-4).  Chunked streaming (oth_welch_accumulate) takes a fresh pilot per chunk. */
#define OTH_DETREND_CONSTANT_EXACT 2 /* = OTH_DETREND_CONSTANT (the name under which the offset-proof detrend was first
                                   asked for; accepted, same builds) */
#define OTH_DETREND_CONSTANT_FAST 3 /* the same operation on the raw samples, -1 ... +2 % of the launch (no pilot, no
                                   subtractions).  Launches of fewer than 8 segments per stream detrend before the
                                   window (as above); the fast builds detrend after the transform, which then carries the
                                   rounding of the DC line m sum(w): about 1e-7 sqrt(nfft / nseg) |m| / sigma of the
                                   detrended power in every bin (measured on MI355X at 2047 segments of 4096 points:
                                   1e-6 at |m| = 30 sigma, 4e-5 at 300 sigma, 1.5e-4 at 3000 sigma), and bins 0, +-1
                                   stand where any float32 mean of the raw samples leaves them, SciPy's included
                                   (2 * 2^-23 |m| |W[k]| / |X[k]|) - inside the 1e-4 parity gate up to a DC line
                                   ~50 dB above the signal's total power. */

/* scaling of the averaged |X|^2 */
#define OTH_SCALE_RAW        0   /* mean over segments of |X|^2 */
#define OTH_SCALE_DENSITY    1   /* / (fs * sum(w^2))   scipy scaling='density' */
#define OTH_SCALE_OVER_N2    2   /* / nfft^2            spectrum_sensor_v2.py:93 */
#define OTH_SCALE_SPECTRUM   3   /* / sum(w)^2          scipy scaling='spectrum' */

/* epilogue of the per-vector periodogram chain */
#define OTH_EPI_MAG          0   /* |X|                 blocks.complex_to_mag, psd_logger.py:53 */
#define OTH_EPI_MAG2         1   /* |X|^2               blocks.complex_to_mag_squared, local_worker.py:65 */
#define OTH_EPI_MAG2_OVER_N2 2   /* |X|^2 / nfft^2      spectrum_sensor_v2.py:92-93 */

/* kernel selection (diagnostics / parity tests) */
#define OTH_KERNEL_AUTO      0
#define OTH_KERNEL_GENERIC   1   /* radix-4 Stockham, powers of two 64 ... 16384 (other lengths: always the any-length kernels) */
#define OTH_KERNEL_TUNED     2   /* register/LDS radix-16 kernels (nfft 256 ... 16384) */

/* how the tuned kernels hand segments to workgroups */
#define OTH_SCHED_CONTIGUOUS  0   /* fixed contiguous runs: bit-reproducible sums */
#define OTH_SCHED_INTERLEAVED 1   /* fixed round-robin chunks: bit-reproducible sums */
#define OTH_SCHED_DYNAMIC     2   /* the plan's initial value = the library's choice: chunks drawn from an atomic
                                     ticket for long 2048 / 4096-point launches (load-balanced; the fp32 summation
                                     order - hence the last bits - may vary run to run), a static schedule where that
                                     measures faster (256 / 512 / 1024 points, whole-segment loads, short launches) */

typedef struct oth_ctx oth_ctx;
typedef struct oth_plan oth_plan;
typedef struct oth_chain oth_chain;

/* ---- library / context ------------------------------------------------- */
int         oth_abi_version(void);
const char *oth_strerror(int code);
int         oth_device_count(int *count);

int         oth_ctx_create(int device_id, oth_ctx **out);
/* adopt an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int         oth_ctx_create_on_stream(int device_id, void *hip_stream, oth_ctx **out);
int         oth_ctx_destroy(oth_ctx *ctx);
const char *oth_last_error(oth_ctx *ctx);
int         oth_ctx_sync(oth_ctx *ctx);
int         oth_ctx_device_name(oth_ctx *ctx, char *buf, size_t buflen);

/* HIP-event timing of the dominant (FFT) kernel on the context's stream.
 * enable != 0 brackets every such launch with events; get() synchronises the
 * stream and returns the sum / count since the last reset. */
int         oth_ctx_set_timing(oth_ctx *ctx, int enable);
int         oth_ctx_get_timing(oth_ctx *ctx, double *total_ms, uint64_t *launches, int reset);

/* ---- device memory helpers (so ctypes-only hosts need no torch) --------- */
int oth_dev_alloc(oth_ctx *ctx, size_t bytes, void **dptr);
int oth_dev_free(oth_ctx *ctx, void *dptr);
int oth_memcpy_h2d(oth_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int oth_memcpy_d2h(oth_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* Synthetic IQ written straight into HBM: unit-power complex
 * AWGN from a counter-based generator + ntones complex exponentials + DC. */
int oth_synth_iq(oth_ctx *ctx, void *iq_dev, size_t nsamples, uint64_t seed, int ntones,
                 const float *tone_amp, const float *tone_freq, float dc_re, float dc_im);

/* Streaming-read probe: a float4 sum over [dptr, dptr+bytes); reports the
 * kernel time so the caller can state the achievable HBM-read peak.  repeats < 0: |repeats| passes of the
 * 8-bytes-per-lane variant (non-temporal float2 loads, the access the FFT kernels use for samples). */
int oth_stream_read_probe(oth_ctx *ctx, const void *dptr, size_t bytes, int repeats, double *ms_per_pass);

/* mean(|x - mean(x)|^2) of a device IQ buffer (Parseval check at full size) */
int oth_iq_power(oth_ctx *ctx, const void *iq_dev, size_t nsamples, double *mean_re, double *mean_im,
                 double *var);

/* ---- Welch PSD ----------------------------------------------------------
 * Replaces scipy.signal.welch as the reference calls it:
 *   ofdm_cr_tools.py:214  (window='flattop', nperseg=nfft)        src_power_welch
 *   ofdm_cr_tools.py:322  (default hann, 50 % overlap)            welch_plot_dB
 *   ofdm_cr_tools.py:342  (same)                                  welch_power_estimate
 *   spectrum_sweeper.py:263 (flattop, nperseg=nfft/4 zero-padded) _src_power
 * followed by the fftshift / excess-bin trim / 10*log10 of
 * spectrum_sweeper.py:265-276.  Two-sided, mean over segments.
 *
 * window: nperseg host floats (NULL = rectangular).  trim_bins bins are dropped
 * from each end AFTER the optional fftshift (np.fft.fftshift: bin k to (k + nfft / 2) mod nfft, odd nfft included);
 * output length = nfft - 2*trim_bins.  nfft: any length, see TRANSFORM LENGTHS above.
 */
int oth_welch_plan(oth_ctx *ctx, int nfft, int nperseg, int noverlap, const float *window,
                   int detrend, int scaling, double fs, int fftshift, int trim_bins, oth_plan **out);
int oth_plan_destroy(oth_plan *plan);
int oth_plan_set_output_db(oth_plan *plan, int enable);      /* 10*log10 in the finalize kernel */
int oth_plan_set_kernel(oth_plan *plan, int which);           /* OTH_KERNEL_* */
int oth_plan_set_schedule(oth_plan *plan, int which);         /* OTH_SCHED_* */
int oth_plan_out_len(oth_plan *plan, int *n);
/* How the blocking / waiting host-output calls of this plan (oth_welch_exec, oth_welch_wait) wait for the GPU (ABI 6; a
 * per-plan setting - round 5 read OTH_HOSTWAIT once per process).  POLL (default): the thread polls the completion word
 * the last launch writes into pinned memory - pause instructions for at most 2 ms, yielding the CPU between looks up to
 * 20 ms, then hipStreamSynchronize; the lowest latency (no interrupt wake-up), at the price of a busy core while it
 * waits.  SYNC: hipStreamSynchronize at once - the mode for a flowgraph with many blocking sensors.  The OTH_HOSTWAIT
 * environment variable ("sync") gives the initial value and is read once, in oth_welch_plan(). */
#define OTH_HOSTWAIT_POLL 0
#define OTH_HOSTWAIT_SYNC 1
int oth_plan_set_hostwait(oth_plan *plan, int mode);
/* Launch tuning for A/B tools and the parity suite: which build of the 4096-point kernel ("dpp", "pipe", "ws") or
 * of the 256 ... 2048-point kernels ("seg3", "seg4": registers held to 3 / 4 waves per SIMD; NULL or "" = the
 * library's choice), or only the detrend form of the size's default kernel ("fd": after the transform even below 8
 * segments per stream of an OTH_DETREND_CONSTANT_FAST plan; "td": before it at any length), the one-role kernel at 8192 points /
 * 50 % overlap instead of the role-split default ("8k1role"), the pilot from its own launch ("plaunch"), a schedule override
 * (-1 = the plan's), segments per chunk and per tail chunk
 * (0 = default).  The OTH_W4096_VARIANT / _SCHED / _CHUNK / _TAIL environment variables give the initial values
 * and are read once, in oth_welch_plan(). */
int oth_plan_set_tuning(oth_plan *plan, const char *variant, int sched, int chunk, int tail_chunk);

/* one-shot: nsamples complex64 -> psd_out[nfft - 2*trim] (host).  Blocking: returns when the PSD is in psd_out.  The
 * last launch writes the row and a completion word into pinned host memory and the call polls that word (no interrupt
 * wake-up; after 20 ms it falls back to a stream synchronisation, which also reports a failed launch;
 * oth_plan_set_hostwait).  Any number of threads may call it on one plan: they run one after the other. */
int oth_welch_exec(oth_plan *plan, const void *iq, size_t nsamples, int src_is_device,
                   float *psd_out, uint64_t *nseg_out);
/* The same step without blocking (ABI 5) - what a gr.sync_block's work() or message handler needs for a Welch scan
 * (reference: python/spectrum_sensor.py:71-75,105-117 -> ofdm_cr_tools.py:471-537; the chain has oth_chain_push_async):
 * _exec_async enqueues copy + kernels and returns a ticket (> 0) at once - the caller's host buffer may be reused when
 * it returns (buffers up to 1 MiB go through a pinned ring and never wait; a larger pageable buffer is staged by the HIP
 * runtime, which may hold the call until earlier work on the stream has finished); _poll looks once (*ready = 0: still running; 1: psd_out / nseg_out are filled); _wait polls until the
 * row is there.  Neither holds the context while it waits.  The plan keeps the last 4 launches: an older ticket
 * reports OTH_ERR_STATE, and _exec_async itself waits only when the GPU is 4 launches behind.  psd_out may be NULL
 * (completion only). */
int oth_welch_exec_async(oth_plan *plan, const void *iq, size_t nsamples, int src_is_device, uint64_t *ticket_out);
int oth_welch_poll(oth_plan *plan, uint64_t ticket, float *psd_out, uint64_t *nseg_out, int *ready);
int oth_welch_wait(oth_plan *plan, uint64_t ticket, float *psd_out, uint64_t *nseg_out);
/* nstreams independent streams laid out every stream_stride samples; device in,
 * device out [nstreams][out_len]; asynchronous. */
int oth_welch_exec_dev(oth_plan *plan, const void *iq_dev, size_t nsamples, int nstreams,
                       size_t stream_stride, float *psd_out_dev, uint64_t *nseg_out);
/* raw sum over segments of |X|^2 (natural bin order, no scale) for time-sharded
 * multi-GPU Welch: partial sums from ranks add, then oth_welch_scale_dev(). */
int oth_welch_partial_dev(oth_plan *plan, const void *iq_dev, size_t nsamples,
                          float *sum_out_dev, uint64_t *nseg_out);
int oth_welch_scale_dev(oth_plan *plan, const float *sum_dev, uint64_t nseg_total, float *psd_out_dev);

/* streaming form used by the sync_block work() host (python/spectrum_sensor.py:71-75
 * contract): chunks of any length; the overlap tail is carried between calls.  accumulate() copies the
 * caller's buffer into a pinned staging slot, enqueues the H2D copy and the kernels, and returns without
 * waiting for the GPU (it waits only if the GPU is still four calls behind); finalize() synchronises. */
int oth_welch_accumulate(oth_plan *plan, const void *iq_host, size_t nsamples);
int oth_welch_finalize(oth_plan *plan, float *psd_out, uint64_t *nseg_out);   /* then resets */
int oth_welch_reset(oth_plan *plan);

/* ---- two-channel cross spectrum / coherence -------
 * Semantics of scipy.signal.csd / coherence with the plan's Welch parameters;
 * produces the first input of coherence_detector (coherence_detector.py:45).
 * Outputs (host, natural FFT order unless the plan has fftshift): pxx, pyy,
 * cxy are float[nfft]; pxy is interleaved re,im float[2*nfft].  Any may be NULL. */
int oth_csd_exec(oth_plan *plan, const void *x, const void *y, size_t nsamples, int src_is_device,
                 float *pxx, float *pyy, float *pxy, float *cxy, uint64_t *nseg_out);

/* Device forms of the same (asynchronous on the context's stream; any output pointer may be NULL). */
int oth_csd_exec_dev(oth_plan *plan, const void *x_dev, const void *y_dev, size_t nsamples, float *pxx_dev,
                     float *pyy_dev, float *pxy_dev, float *cxy_dev, uint64_t *nseg_out);
/* Raw sums over segments for time-sharded multi-GPU coherence: sums_out_dev is
 * float[4 * nfft] = sum |X|^2 [nfft], sum |Y|^2 [nfft], sum conj(X) Y [nfft] interleaved re,im; natural bin order,
 * no scale / shift / trim.  Partial sums of ranks add; oth_csd_scale_dev() then applies the plan's scaling for
 * nseg_total segments, its fftshift / trim, and forms Cxy = |Pxy|^2 / (Pxx Pyy). */
int oth_csd_partial_dev(oth_plan *plan, const void *x_dev, const void *y_dev, size_t nsamples, float *sums_out_dev,
                        uint64_t *nseg_out);
int oth_csd_scale_dev(oth_plan *plan, const float *sums_dev, uint64_t nseg_total, float *pxx_dev, float *pyy_dev,
                      float *pxy_dev, float *cxy_dev);

/* ---- per-vector periodogram chain ----------------------------------------
 * Replaces the GNU Radio chain
 *   stream_to_vector -> keep_one_in_n -> fft_vcc(N, True, window, shift) ->
 *   complex_to_mag[_squared] [-> multiply_const(1/N^2)]
 *   [-> single_pole_iir_filter_ff -> nlog10_ff]
 * of spectrum_sensor_v2.py:85-93, psd_logger.py:43-53, local_worker.py:58-69,
 * multichannel_scanner.py:78-86.  The chain keeps GNU Radio's stream state
 * between calls: leftover samples of a partial vector, the keep_one_in_n
 * counter, the IIR memory and the peak-hold vector.  nfft: any length (TRANSFORM LENGTHS above).
 */
int oth_chain_create(oth_ctx *ctx, int nfft, const float *window, int fftshift, int epilogue,
                     int keep_one_in_n, oth_chain **out);
int oth_chain_destroy(oth_chain *chain);
int oth_chain_set_keep_one_in_n(oth_chain *chain, int n);     /* local_worker.py:85-87 set_rate */
/* single_pole_iir_filter_ff(alpha) + nlog10_ff(10, N, k_db); alpha<=0 disables */
int oth_chain_set_iir_log(oth_chain *chain, float alpha, float k_db);
int oth_chain_set_peak_hold(oth_chain *chain, int enable);    /* psd_logger.py:85 */
int oth_chain_set_kernel(oth_chain *chain, int which);         /* OTH_KERNEL_GENERIC: coverage kernels (parity tests) */
int oth_chain_reset(oth_chain *chain);
/* feed nsamples (host, or device when src_is_device); rows_out (host, may be
 * NULL) receives up to rows_capacity post-epilogue rows (dB rows when the IIR/log
 * stage is on); nrows_out = rows produced by this call. */
int oth_chain_push(oth_chain *chain, const void *iq, size_t nsamples, int src_is_device,
                   float *rows_out, size_t rows_capacity, uint64_t *nrows_out);
/* device in, device out ([rows_capacity][nfft], may be NULL), asynchronous, never synchronises */
int oth_chain_push_dev(oth_chain *chain, const void *iq_dev, size_t nsamples, float *rows_out_dev,
                       size_t rows_capacity, uint64_t *nrows_out);
/* The sync_block.work() form (python/spectrum_sensor.py:71-75: must not block; input valid only during the
 * call; consumers behind message_sink(dont_block) + msg_queue(2) see the latest vector,
 * spectrum_sensor_v2.py:71-72,97,404-414): the samples are copied into a pinned ring slot, the H2D copy + the kernels
 * are enqueued - the closing kernel writes the LATEST row straight into the slot's pinned host row - an event is
 * recorded and the call returns a ticket.  A push none of whose vectors survives keep_one_in_n enqueues nothing at all.  poll() is
 * non-blocking (ready = 0 while the GPU is still working); wait() blocks without holding the context.  The ring
 * keeps the last four tickets: an older one returns OTH_ERR_STATE (it lost against newer vectors). */
/* (Pushes above 1 MiB of PAGEABLE host memory skip the pinned slot and use the runtime's staged copy, which returns once
 * the caller's buffer has been read but may hold the host until the stream reaches the copy; GNU Radio's work() chunks
 * are far smaller.) */
int oth_chain_push_async(oth_chain *chain, const void *iq_host, size_t nsamples, uint64_t *ticket_out);
int oth_chain_poll(oth_chain *chain, uint64_t ticket, float *row_out, uint64_t *nrows_out, int *ready);
int oth_chain_wait(oth_chain *chain, uint64_t ticket, float *row_out, uint64_t *nrows_out);
/* rows the push behind `ticket` produces (known when it is enqueued; never waits): lets a consumer that counts
 * vectors - the waterfall's keep_one_in_n(sens_per_sec), spectrum_sensor_v2.py:102 - count the ones it drops too */
int oth_chain_ticket_rows(oth_chain *chain, uint64_t ticket, uint64_t *nrows_out);
/* Stream operations (asynchronous copies + kernel launches; event records not counted) the LAST push of this chain
 * enqueued (ABI 6) - what a work()-sized push costs the scheduler thread: 0 for a push all of whose vectors keep_one_in_n
 * drops (nothing is copied or launched; the ticket is ready at once), 2 for a chain without state (H2D + one kernel that
 * writes the latest row into pinned host memory), 3 with the IIR / peak-hold state (tests/test_blocks_gpu.py). */
int oth_chain_last_push_ops(oth_chain *chain, uint64_t *ops_out);
int oth_chain_get_peak(oth_chain *chain, float *peak_out);    /* float[nfft] */
int oth_chain_get_iir(oth_chain *chain, float *lin_out);      /* float[nfft], linear IIR state */
/* mean of each `group` consecutive rows (BASELINE config 1 "8-seg avg") */
int oth_rows_group_mean(oth_ctx *ctx, const float *rows_host, size_t nrows, int nfft, int group,
                        float *out_host);

/* ---- channel power -------------------------------------------------------
 * src_power (ofdm_cr_tools.py:232-249): |convolve(psd, ones(int(sb))/sb, 'same')|
 * then per-channel slice sums.  lo/hi are the slice bounds the host computed with
 * the reference's int() arithmetic; power_out[nch]; also returns min power. */
int oth_channel_power(oth_ctx *ctx, const float *psd_host, int nfft, double srch_bins, int nch,
                      const int *lo, const int *hi, float *power_out, float *movavg_out /*nullable*/);

/* Per-bin energy detection for the batched scanner (BASELINE config 5; the per-bin analogue of the
 * channel threshold of spectrum_sensor_v2.py:465-477): noise = min_k movingaverage(psd)[k],
 * mask[k] = psd[k] > thr_leveler * noise.  nrows PSD rows of nfft bins each (host); mask_out is
 * uint8[nrows][nfft], noise_out float[nrows] (nullable). */
int oth_bin_threshold(oth_ctx *ctx, const float *psd_host, int nrows, int nfft, double srch_bins,
                      float thr_leveler, unsigned char *mask_out, float *noise_out);

/* Decision stage of the batched scanner on PSD rows that are already in HBM (oth_welch_exec_dev with nstreams
 * rows): moving average once per row, channel slice sums (src_power, ofdm_cr_tools.py:232-249), noise floor and
 * per-bin mask (as oth_bin_threshold) in one launch sequence on context-owned scratch - no copy of the rows, no
 * allocation per call.  Host outputs: mask_out uint8[nrows][nfft] (nullable), noise_out float[nrows] (nullable),
 * power_out float[nrows][nch] (required when nch > 0). */
int oth_scan_decide_dev(oth_ctx *ctx, const float *psd_rows_dev, int nrows, int nfft, double srch_bins,
                        float thr_leveler, int nch, const int *lo, const int *hi, unsigned char *mask_out,
                        float *noise_out, float *power_out);
/* the same stage with DEVICE outputs (asynchronous on the context's stream, no host copy of anything but the
 * channel slice bounds): mask_dev [nrows][nfft] bytes (nullable), noise_dev [nrows], power_dev [nrows][nch]
 * (nullable when nch == 0).  The sharded scanner (multichannel_scanner over ranks) feeds these
 * straight into its all-gather. */
int oth_scan_decide_dev_out(oth_ctx *ctx, const float *psd_rows_dev, int nrows, int nfft, double srch_bins,
                            float thr_leveler, int nch, const int *lo, const int *hi, unsigned char *mask_dev,
                            float *noise_dev, float *power_dev);

/* ---- xcorr (ofdm_cr_tools.py:155-161) ------------------------------------
 * |fftshift(ifft(fft(b,L) * conj(fft(a,L))))[L/2:]| for any L (TRANSFORM LENGTHS above; `L/2` is the reference's
 * Python-2 integer division: L - L/2 outputs); a, b host complex64 of na, nb samples, zero-padded to L or, like
 * np.fft.fft(a, L), cut to their first L; out float[L - L/2]. */
int oth_xcorr(oth_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, int L, float *out);
/* fac (ofdm_cr_tools.py:163-166): |fftshift(fft(|fft(data,L)|, L))[L/2:]| */
int oth_fac(oth_ctx *ctx, const void *data, size_t n, int L, float *out);

/* ---- diagnostics (ABI 5) ------------------------------------------------------------------------------------------
 * Which kernel build, detrend form, pilot, schedule, chunk sizes, grid and partial-row layout a launch takes is decided by
 * pure host logic (csrc/api.hip resolve_recipe) and can be read back as text:
 *   "kernel=welch4096:ws nfft=4096 form=freq pilot=inline sched=dynamic chunk=20 tail=5 nbig=6297 bpc=2 W=512 rows=1 nch=1 layout=1"
 * oth__debug_recipe needs NO device: window_class 0 = all ones, 1 = confined spectrum (periodic cosine-sum windows),
 * 2 = wide (no detrend table), 3 = confined to 256 nfft / 4096 bins only; runtime_occupancy 0 = resident workgroups per CU
 * from the built-in MI355X table, 1 = from the occupancy calculator (needs a GPU).  oth__debug_last_recipe: the recipe of
 * the plan's last averaging launch. */
int oth__debug_recipe(int nfft, int nperseg, int noverlap, int window_class, int detrend_mode, int two_channel, int kernel_pref,
                      const char *variant, int sched_pref, long long nseg, int nstreams, int cu_count, int runtime_occupancy,
                      char *buf, size_t buflen);
int oth__debug_last_recipe(oth_plan *plan, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* OFDM_TOOLS_HIP_H */
