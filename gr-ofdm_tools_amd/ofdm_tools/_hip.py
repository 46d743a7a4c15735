"""ctypes binding of libofdmtools_hip.so (include/ofdm_tools_hip.h).

The only compute back end of this package.  There is no NumPy/SciPy fallback:
if the shared library is missing or no MI355X is visible every entry point
raises :class:`HipUnavailable` / :class:`HipError` with the reason.
"""
import ctypes as C
import os
import threading

import numpy as np

OK = 0
DETREND_NONE, DETREND_CONSTANT, DETREND_CONSTANT_EXACT, DETREND_CONSTANT_FAST = 0, 1, 2, 3
SCALE_RAW, SCALE_DENSITY, SCALE_OVER_N2, SCALE_SPECTRUM = 0, 1, 2, 3
EPI_MAG, EPI_MAG2, EPI_MAG2_OVER_N2 = 0, 1, 2
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_TUNED = 0, 1, 2
SCHED_CONTIGUOUS, SCHED_INTERLEAVED, SCHED_DYNAMIC = 0, 1, 2
HOSTWAIT_POLL, HOSTWAIT_SYNC = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OFDM_TOOLS_HIP_LIB',
                          os.path.join(os.path.dirname(_HERE), 'lib', 'libofdmtools_hip.so'))


class HipUnavailable(RuntimeError):
    """libofdmtools_hip.so could not be loaded (not built, or ROCm runtime missing)."""


class HipError(RuntimeError):
    def __init__(self, code, where, detail):
        RuntimeError.__init__(self, '%s failed (%d): %s' % (where, code, detail))
        self.code = code


_p = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_f = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); must list every symbol include/ofdm_tools_hip.h declares
SIGNATURES = {
    'oth_abi_version': (C.c_int, []),
    'oth_strerror': (C.c_char_p, [C.c_int]),
    'oth_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'oth_ctx_create': (C.c_int, [C.c_int, _pp]),
    'oth_ctx_create_on_stream': (C.c_int, [C.c_int, _p, _pp]),
    'oth_ctx_destroy': (C.c_int, [_p]),
    'oth_last_error': (C.c_char_p, [_p]),
    'oth_ctx_sync': (C.c_int, [_p]),
    'oth_ctx_device_name': (C.c_int, [_p, C.c_char_p, C.c_size_t]),
    'oth_ctx_set_timing': (C.c_int, [_p, C.c_int]),
    'oth_ctx_get_timing': (C.c_int, [_p, C.POINTER(C.c_double), _u64p, C.c_int]),
    'oth_dev_alloc': (C.c_int, [_p, C.c_size_t, _pp]),
    'oth_dev_free': (C.c_int, [_p, _p]),
    'oth_memcpy_h2d': (C.c_int, [_p, _p, _p, C.c_size_t]),
    'oth_memcpy_d2h': (C.c_int, [_p, _p, _p, C.c_size_t]),
    'oth_synth_iq': (C.c_int, [_p, _p, C.c_size_t, C.c_uint64, C.c_int, _f, _f, C.c_float, C.c_float]),
    'oth_stream_read_probe': (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    'oth_iq_power': (C.c_int, [_p, _p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]),
    'oth_welch_plan': (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.c_int, _pp]),
    'oth_plan_destroy': (C.c_int, [_p]),
    'oth_plan_set_output_db': (C.c_int, [_p, C.c_int]),
    'oth_plan_set_kernel': (C.c_int, [_p, C.c_int]),
    'oth_plan_set_schedule': (C.c_int, [_p, C.c_int]),
    'oth_plan_out_len': (C.c_int, [_p, C.POINTER(C.c_int)]),
    'oth_plan_set_hostwait': (C.c_int, [_p, C.c_int]),
    'oth_plan_set_tuning': (C.c_int, [_p, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    'oth_welch_exec': (C.c_int, [_p, _p, C.c_size_t, C.c_int, _f, _u64p]),
    'oth_welch_exec_async': (C.c_int, [_p, _p, C.c_size_t, C.c_int, _u64p]),
    'oth_welch_poll': (C.c_int, [_p, C.c_uint64, _f, _u64p, C.POINTER(C.c_int)]),
    'oth_welch_wait': (C.c_int, [_p, C.c_uint64, _f, _u64p]),
    'oth_welch_exec_dev': (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_size_t, _p, _u64p]),
    'oth_welch_partial_dev': (C.c_int, [_p, _p, C.c_size_t, _p, _u64p]),
    'oth_welch_scale_dev': (C.c_int, [_p, _p, C.c_uint64, _p]),
    'oth_welch_accumulate': (C.c_int, [_p, _p, C.c_size_t]),
    'oth_welch_finalize': (C.c_int, [_p, _f, _u64p]),
    'oth_welch_reset': (C.c_int, [_p]),
    'oth_csd_exec': (C.c_int, [_p, _p, _p, C.c_size_t, C.c_int, _f, _f, _f, _f, _u64p]),
    'oth_csd_exec_dev': (C.c_int, [_p, _p, _p, C.c_size_t, _p, _p, _p, _p, _u64p]),
    'oth_csd_partial_dev': (C.c_int, [_p, _p, _p, C.c_size_t, _p, _u64p]),
    'oth_csd_scale_dev': (C.c_int, [_p, _p, C.c_uint64, _p, _p, _p, _p]),
    'oth_chain_create': (C.c_int, [_p, C.c_int, _f, C.c_int, C.c_int, C.c_int, _pp]),
    'oth_chain_destroy': (C.c_int, [_p]),
    'oth_chain_set_keep_one_in_n': (C.c_int, [_p, C.c_int]),
    'oth_chain_set_iir_log': (C.c_int, [_p, C.c_float, C.c_float]),
    'oth_chain_set_peak_hold': (C.c_int, [_p, C.c_int]),
    'oth_chain_set_kernel': (C.c_int, [_p, C.c_int]),
    'oth_chain_reset': (C.c_int, [_p]),
    'oth_chain_push': (C.c_int, [_p, _p, C.c_size_t, C.c_int, _f, C.c_size_t, _u64p]),
    'oth_chain_push_dev': (C.c_int, [_p, _p, C.c_size_t, _p, C.c_size_t, _u64p]),
    'oth_chain_push_async': (C.c_int, [_p, _p, C.c_size_t, _u64p]),
    'oth_chain_poll': (C.c_int, [_p, C.c_uint64, _f, _u64p, C.POINTER(C.c_int)]),
    'oth_chain_wait': (C.c_int, [_p, C.c_uint64, _f, _u64p]),
    'oth_chain_ticket_rows': (C.c_int, [_p, C.c_uint64, _u64p]),
    'oth_chain_last_push_ops': (C.c_int, [_p, _u64p]),
    'oth_chain_get_peak': (C.c_int, [_p, _f]),
    'oth_chain_get_iir': (C.c_int, [_p, _f]),
    'oth_rows_group_mean': (C.c_int, [_p, _f, C.c_size_t, C.c_int, C.c_int, _f]),
    'oth_channel_power': (C.c_int, [_p, _f, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    _f, _f]),
    'oth_bin_threshold': (C.c_int, [_p, _f, C.c_int, C.c_int, C.c_double, C.c_float, C.POINTER(C.c_ubyte), _f]),
    'oth_scan_decide_dev': (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_double, C.c_float, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_ubyte), _f, _f]),
    'oth_scan_decide_dev_out': (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_double, C.c_float, C.c_int, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), _p, _p, _p]),
    'oth_xcorr': (C.c_int, [_p, _p, C.c_size_t, _p, C.c_size_t, C.c_int, _f]),
    'oth_fac': (C.c_int, [_p, _p, C.c_size_t, C.c_int, _f]),
    'oth__debug_recipe': (C.c_int, [C.c_int] * 7 + [C.c_char_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                                    C.c_size_t]),
    'oth__debug_last_recipe': (C.c_int, [_p, C.c_char_p, C.c_size_t]),
}

_lib = None
_lib_lock = threading.Lock()


def load():
    """Load the shared library once and attach the prototypes."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and finds no GPU if another copy
        # (the system ROCm's, which this library would pull in) was loaded first.  Multi-GPU hosts use torch for
        # device tensors and torch.distributed, so when torch is installed its runtime goes in first and this
        # library binds to it (same SONAME).  OFDM_TOOLS_HIP_STANDALONE=1 skips the import.
        if os.environ.get('OFDM_TOOLS_HIP_STANDALONE') != '1':
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable('%s not found - build it with `make -C gr-ofdm_tools_amd` '
                                 '(or __graft_entry__.build()); this package has no CPU fallback' % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipUnavailable('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _c64(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    if x.ndim != 1:
        x = x.reshape(-1)
    return x


def _fptr(a):
    return a.ctypes.data_as(_f)


class Context(object):
    """One device + one HIP stream (oth_ctx).  The library serialises calls per context, so blocks
    running on different scheduler threads may share one."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = C.c_void_p()
        if stream is None:
            rc = self.lib.oth_ctx_create(int(device), C.byref(h))
        else:
            rc = self.lib.oth_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc != OK:
            raise HipError(rc, 'oth_ctx_create', self.lib.oth_last_error(None).decode())
        self.h = h
        self.device = int(device)
        self.stream = None if stream is None else int(stream)      # the adopted hipStream_t, if any
        self._plans_lock = threading.Lock()

    def on_torch_stream(self):
        """True when this context runs on torch's current stream of its device: kernels and torch ops
        (collectives included) are then ordered by the stream and need no host synchronisation between them."""
        if self.stream is None:
            return False
        import torch
        return int(torch.cuda.current_stream(self.device).cuda_stream) == self.stream

    def check(self, rc, where):
        if rc != OK:
            raise HipError(rc, where, self.lib.oth_last_error(self.h).decode())

    def close(self):
        if getattr(self, 'h', None):
            for plan in self.__dict__.pop('_plans', {}).values():
                plan.close()
            self.lib.oth_ctx_destroy(self.h)
            self.h = None

    def cached_plan(self, key, make, limit=64):
        """One plan per (context, key) for callers that ask for the same shape with every request (the legacy helpers of
        ofdm_cr_tools): a plan owns device tables and scratch, and building one costs allocations and a stream
        synchronisation.  GNU Radio block threads share the default context, so the cache is locked; a hit moves to the
        young end (least-recently-used eviction), and past `limit` shapes the oldest plan WITHOUT an uncollected
        exec_async() ticket goes - a plan that still owes a result is kept even if that overshoots the limit (closing it
        would turn the ticket's poll / wait into an error inside work()).  The cache is closed with the context."""
        with self._plans_lock:
            plans = self.__dict__.setdefault('_plans', {})
            plan = plans.pop(key, None)
            if plan is None or not plan.h:
                plan = make()
            plans[key] = plan                  # (re-)inserted at the young end
            while len(plans) > limit:
                victim = next((k for k, v in plans.items() if k != key and not getattr(v, 'outstanding', 0)), None)
                if victim is None:
                    break
                plans.pop(victim).close()
            return plan

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -------------------------------------------------------------
    def sync(self):
        self.check(self.lib.oth_ctx_sync(self.h), 'oth_ctx_sync')

    def device_name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.oth_ctx_device_name(self.h, buf, 256), 'oth_ctx_device_name')
        return buf.value.decode()

    def set_timing(self, on):
        self.check(self.lib.oth_ctx_set_timing(self.h, 1 if on else 0), 'oth_ctx_set_timing')

    def get_timing(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        self.check(self.lib.oth_ctx_get_timing(self.h, C.byref(ms), C.byref(n), 1 if reset else 0),
                   'oth_ctx_get_timing')
        return ms.value, n.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.oth_dev_alloc(self.h, nbytes, C.byref(p)), 'oth_dev_alloc')
        return p.value

    def free(self, ptr):
        self.check(self.lib.oth_dev_free(self.h, C.c_void_p(ptr)), 'oth_dev_free')

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.oth_memcpy_h2d(self.h, C.c_void_p(dptr), arr.ctypes.data_as(_p), arr.nbytes),
                   'oth_memcpy_h2d')

    def d2h(self, dptr, shape, dtype):
        out = np.empty(shape, dtype)
        self.check(self.lib.oth_memcpy_d2h(self.h, out.ctypes.data_as(_p), C.c_void_p(dptr), out.nbytes),
                   'oth_memcpy_d2h')
        return out

    def synth_iq(self, dptr, nsamples, seed, tones=(), dc=0j):
        amp = np.array([t[0] for t in tones], np.float32)
        frq = np.array([t[1] for t in tones], np.float32)
        self.check(self.lib.oth_synth_iq(self.h, C.c_void_p(dptr), nsamples, seed, len(tones),
                                         _fptr(amp) if len(tones) else None, _fptr(frq) if len(tones) else None,
                                         float(np.real(dc)), float(np.imag(dc))), 'oth_synth_iq')

    def stream_read_probe(self, dptr, nbytes, repeats=5):
        ms = C.c_double()
        self.check(self.lib.oth_stream_read_probe(self.h, C.c_void_p(dptr), nbytes, repeats, C.byref(ms)),
                   'oth_stream_read_probe')
        return ms.value

    def iq_power(self, dptr, nsamples):
        mr, mi, var = C.c_double(), C.c_double(), C.c_double()
        self.check(self.lib.oth_iq_power(self.h, C.c_void_p(dptr), nsamples, C.byref(mr), C.byref(mi),
                                         C.byref(var)), 'oth_iq_power')
        return complex(mr.value, mi.value), var.value

    # -- factories ------------------------------------------------------------
    def welch_plan(self, nfft, nperseg=None, noverlap=None, window=None, detrend=DETREND_CONSTANT,
                   scaling=SCALE_DENSITY, fs=1.0, fftshift=False, trim_bins=0, db=False, kernel=KERNEL_AUTO):
        return WelchPlan(self, nfft, nperseg, noverlap, window, detrend, scaling, fs, fftshift, trim_bins, db,
                         kernel)

    def chain(self, nfft, window=None, fftshift=True, epilogue=EPI_MAG2, keep_one_in_n=1):
        return Chain(self, nfft, window, fftshift, epilogue, keep_one_in_n)

    # -- small ops --------------------------------------------------------------
    def rows_group_mean(self, rows, group):
        rows = np.ascontiguousarray(rows, np.float32)
        nrows, nfft = rows.shape
        out = np.empty((nrows // group, nfft), np.float32)
        self.check(self.lib.oth_rows_group_mean(self.h, _fptr(rows), nrows, nfft, group, _fptr(out)),
                   'oth_rows_group_mean')
        return out

    def channel_power(self, psd, srch_bins, lo, hi, want_movavg=False):
        psd = np.ascontiguousarray(psd, np.float32)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        out = np.empty(len(lo), np.float32)
        ma = np.empty(len(psd), np.float32) if want_movavg else None
        self.check(self.lib.oth_channel_power(self.h, _fptr(psd), len(psd), float(srch_bins), len(lo),
                                              lo.ctypes.data_as(C.POINTER(C.c_int)),
                                              hi.ctypes.data_as(C.POINTER(C.c_int)), _fptr(out),
                                              _fptr(ma) if want_movavg else None), 'oth_channel_power')
        return (out, ma) if want_movavg else out

    def bin_threshold(self, psd_rows, srch_bins, thr_leveler):
        """-> (mask uint8[nrows][nfft], noise float32[nrows]) for PSD rows (2-D) or one row (1-D)."""
        rows = np.ascontiguousarray(np.atleast_2d(psd_rows), np.float32)
        nrows, nfft = rows.shape
        mask = np.empty((nrows, nfft), np.uint8)
        noise = np.empty(nrows, np.float32)
        self.check(self.lib.oth_bin_threshold(self.h, _fptr(rows), nrows, nfft, float(srch_bins), float(thr_leveler),
                                              mask.ctypes.data_as(C.POINTER(C.c_ubyte)), _fptr(noise)),
                   'oth_bin_threshold')
        return mask, noise

    def scan_decide_dev(self, rows_dptr, nrows, nfft, srch_bins, thr_leveler, lo=(), hi=(), want_mask=True):
        """Decision stage on device-resident PSD rows -> (mask uint8[nrows][nfft] or None, noise[nrows],
        power[nrows][nch])."""
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        nch = len(lo)
        mask = np.empty((nrows, nfft), np.uint8) if want_mask else None
        noise = np.empty(nrows, np.float32)
        power = np.empty((nrows, nch), np.float32)
        ip = C.POINTER(C.c_int)
        self.check(self.lib.oth_scan_decide_dev(self.h, C.c_void_p(rows_dptr), int(nrows), int(nfft), float(srch_bins),
                                                float(thr_leveler), nch, lo.ctypes.data_as(ip) if nch else None,
                                                hi.ctypes.data_as(ip) if nch else None,
                                                mask.ctypes.data_as(C.POINTER(C.c_ubyte)) if want_mask else None,
                                                _fptr(noise), _fptr(power) if nch else None), 'oth_scan_decide_dev')
        return mask, noise, power

    def scan_decide_dev_out(self, rows_dptr, nrows, nfft, srch_bins, thr_leveler, lo, hi, noise_dptr, power_dptr,
                            mask_dptr=0):
        """The same stage with device outputs (asynchronous): noise[nrows], power[nrows][len(lo)], optional mask."""
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        nch = len(lo)
        ip = C.POINTER(C.c_int)
        self.check(self.lib.oth_scan_decide_dev_out(self.h, C.c_void_p(rows_dptr), int(nrows), int(nfft), float(srch_bins),
                                                    float(thr_leveler), nch, lo.ctypes.data_as(ip) if nch else None,
                                                    hi.ctypes.data_as(ip) if nch else None,
                                                    C.c_void_p(mask_dptr) if mask_dptr else None, C.c_void_p(noise_dptr),
                                                    C.c_void_p(power_dptr) if nch else None), 'oth_scan_decide_dev_out')

    def xcorr(self, a, b, length):
        a, b = _c64(a)[:length], _c64(b)[:length]
        out = np.empty(length - length // 2, np.float32)
        self.check(self.lib.oth_xcorr(self.h, a.ctypes.data_as(_p), len(a), b.ctypes.data_as(_p), len(b),
                                      int(length), _fptr(out)), 'oth_xcorr')
        return out

    def fac(self, data, length):
        a = _c64(data)[:length]
        out = np.empty(length - length // 2, np.float32)
        self.check(self.lib.oth_fac(self.h, a.ctypes.data_as(_p), len(a), int(length), _fptr(out)), 'oth_fac')
        return out


class WelchPlan(object):
    def __init__(self, ctx, nfft, nperseg, noverlap, window, detrend, scaling, fs, fftshift, trim_bins, db, kernel):
        self.ctx = ctx
        nperseg = int(nfft if nperseg is None else nperseg)
        noverlap = int(nperseg // 2 if noverlap is None else noverlap)
        self.nfft, self.nperseg, self.noverlap = int(nfft), nperseg, noverlap
        self.step = nperseg - noverlap
        w = None
        if window is not None:
            w = np.ascontiguousarray(window, np.float32)
            if w.shape != (nperseg,):
                raise ValueError('window must have nperseg=%d entries' % nperseg)
        h = C.c_void_p()
        ctx.check(ctx.lib.oth_welch_plan(ctx.h, int(nfft), nperseg, noverlap, _fptr(w) if w is not None else None,
                                         int(detrend), int(scaling), float(fs), 1 if fftshift else 0,
                                         int(trim_bins), C.byref(h)), 'oth_welch_plan')
        self.h = h
        n = C.c_int()
        ctx.check(ctx.lib.oth_plan_out_len(h, C.byref(n)), 'oth_plan_out_len')
        self.out_len = n.value
        if db:
            ctx.check(ctx.lib.oth_plan_set_output_db(h, 1), 'oth_plan_set_output_db')
        if kernel != KERNEL_AUTO:
            ctx.check(ctx.lib.oth_plan_set_kernel(h, int(kernel)), 'oth_plan_set_kernel')

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.oth_plan_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, which):
        self.ctx.check(self.ctx.lib.oth_plan_set_kernel(self.h, int(which)), 'oth_plan_set_kernel')

    def set_tuning(self, variant=None, sched=-1, chunk=0, tail=0):
        """A/B tools and the parity suite: pick a build of the 4096-point kernel ('dpp', 'pipe', 'ws'), override the
        schedule, set the segments per chunk / tail chunk.  Defaults restore the library's choices."""
        v = variant.encode() if variant else None
        self.ctx.check(self.ctx.lib.oth_plan_set_tuning(self.h, v, int(sched), int(chunk), int(tail)),
                       'oth_plan_set_tuning')

    def set_schedule(self, which):
        self.ctx.check(self.ctx.lib.oth_plan_set_schedule(self.h, int(which)), 'oth_plan_set_schedule')

    def set_hostwait(self, sync):
        """True: exec() / wait() sleep in hipStreamSynchronize instead of polling the completion word (low CPU)."""
        self.ctx.check(self.ctx.lib.oth_plan_set_hostwait(self.h, HOSTWAIT_SYNC if sync else HOSTWAIT_POLL), 'oth_plan_set_hostwait')

    def nseg(self, nsamples):
        return (nsamples - self.noverlap) // self.step if nsamples >= self.nperseg else 0

    def last_recipe(self):
        """Diagnostics: 'kernel=... form=... pilot=... sched=... chunk=... W=...' of the last averaging launch."""
        buf = C.create_string_buffer(512)
        self.ctx.check(self.ctx.lib.oth__debug_last_recipe(self.h, buf, 512), 'oth__debug_last_recipe')
        return buf.value.decode()

    def exec(self, x):
        """x: host complex64 array -> float32 PSD of out_len bins."""
        x = _c64(x)
        out = np.empty(self.out_len, np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_welch_exec(self.h, x.ctypes.data_as(_p), len(x), 0, _fptr(out),
                                                   C.byref(n)), 'oth_welch_exec')
        self.last_nseg = n.value
        return out

    def exec_device_src(self, dptr, nsamples):
        out = np.empty(self.out_len, np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_welch_exec(self.h, C.c_void_p(dptr), nsamples, 1, _fptr(out), C.byref(n)),
                       'oth_welch_exec')
        self.last_nseg = n.value
        return out

    def exec_async(self, x, nsamples=None):
        """work() form: enqueue one Welch scan and return a ticket at once.  x: host complex64 array (may be reused as
        soon as the call returns), or a device pointer when nsamples is given."""
        t = C.c_uint64()
        if nsamples is None:
            x = _c64(x)
            rc = self.ctx.lib.oth_welch_exec_async(self.h, x.ctypes.data_as(_p), len(x), 0, C.byref(t))
        else:
            rc = self.ctx.lib.oth_welch_exec_async(self.h, C.c_void_p(x), int(nsamples), 1, C.byref(t))
        self.ctx.check(rc, 'oth_welch_exec_async')
        self.outstanding = getattr(self, 'outstanding', 0) + 1      # tickets not collected yet (Context.cached_plan keeps such a plan)
        return int(t.value)

    def poll(self, ticket):
        """-> None while the GPU is still working, else the float32 PSD (last_nseg is set)."""
        out = np.empty(self.out_len, np.float32)
        n, ready = C.c_uint64(), C.c_int()
        self.ctx.check(self.ctx.lib.oth_welch_poll(self.h, int(ticket), _fptr(out), C.byref(n), C.byref(ready)),
                       'oth_welch_poll')
        if not ready.value:
            return None
        self.outstanding = max(0, getattr(self, 'outstanding', 0) - 1)
        self.last_nseg = n.value
        return out

    def wait(self, ticket):
        out = np.empty(self.out_len, np.float32)
        n = C.c_uint64()
        try:
            self.ctx.check(self.ctx.lib.oth_welch_wait(self.h, int(ticket), _fptr(out), C.byref(n)), 'oth_welch_wait')
        finally:
            self.outstanding = max(0, getattr(self, 'outstanding', 0) - 1)
        self.last_nseg = n.value
        return out

    def exec_dev(self, dptr, nsamples, out_dptr, nstreams=1, stream_stride=None):
        """Asynchronous: device in, device out ([nstreams][out_len] float32)."""
        n = C.c_uint64()
        stride = nsamples if stream_stride is None else stream_stride
        self.ctx.check(self.ctx.lib.oth_welch_exec_dev(self.h, C.c_void_p(dptr), nsamples, nstreams, stride,
                                                       C.c_void_p(out_dptr), C.byref(n)), 'oth_welch_exec_dev')
        return n.value

    def partial_dev(self, dptr, nsamples, sum_dptr):
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_welch_partial_dev(self.h, C.c_void_p(dptr), nsamples, C.c_void_p(sum_dptr),
                                                          C.byref(n)), 'oth_welch_partial_dev')
        return n.value

    def scale_dev(self, sum_dptr, nseg_total, out_dptr):
        self.ctx.check(self.ctx.lib.oth_welch_scale_dev(self.h, C.c_void_p(sum_dptr), nseg_total,
                                                        C.c_void_p(out_dptr)), 'oth_welch_scale_dev')

    def accumulate(self, x):
        x = _c64(x)
        self.ctx.check(self.ctx.lib.oth_welch_accumulate(self.h, x.ctypes.data_as(_p), len(x)),
                       'oth_welch_accumulate')

    def finalize(self):
        out = np.empty(self.out_len, np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_welch_finalize(self.h, _fptr(out), C.byref(n)), 'oth_welch_finalize')
        self.last_nseg = n.value
        return out

    def reset(self):
        self.ctx.check(self.ctx.lib.oth_welch_reset(self.h), 'oth_welch_reset')

    def csd(self, x, y):
        """-> pxx, pyy, pxy (complex64), cxy for host inputs."""
        x, y = _c64(x), _c64(y)
        if len(x) != len(y):
            raise ValueError('x and y must have the same length')
        m = self.out_len
        pxx, pyy, cxy = (np.empty(m, np.float32) for _ in range(3))
        pxy = np.empty(2 * m, np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_csd_exec(self.h, x.ctypes.data_as(_p), y.ctypes.data_as(_p), len(x), 0,
                                                 _fptr(pxx), _fptr(pyy), _fptr(pxy), _fptr(cxy), C.byref(n)),
                       'oth_csd_exec')
        self.last_nseg = n.value
        return pxx, pyy, pxy.view(np.complex64), cxy


def _csd_device_src(self, dx, dy, nsamples):
    """-> pxx, pyy, pxy (complex64), cxy for device inputs."""
    m = self.out_len
    pxx, pyy, cxy = (np.empty(m, np.float32) for _ in range(3))
    pxy = np.empty(2 * m, np.float32)
    n = C.c_uint64()
    self.ctx.check(self.ctx.lib.oth_csd_exec(self.h, C.c_void_p(dx), C.c_void_p(dy), nsamples, 1, _fptr(pxx),
                                             _fptr(pyy), _fptr(pxy), _fptr(cxy), C.byref(n)), 'oth_csd_exec')
    self.last_nseg = n.value
    return pxx, pyy, pxy.view(np.complex64), cxy


WelchPlan.csd_device_src = _csd_device_src


def _csd_exec_dev(self, dx, dy, nsamples, pxx=0, pyy=0, pxy=0, cxy=0):
    """Asynchronous: device in, device out (any output pointer may be 0)."""
    n = C.c_uint64()
    vp = lambda v: C.c_void_p(v) if v else None      # noqa: E731
    self.ctx.check(self.ctx.lib.oth_csd_exec_dev(self.h, C.c_void_p(dx), C.c_void_p(dy), nsamples, vp(pxx), vp(pyy),
                                                 vp(pxy), vp(cxy), C.byref(n)), 'oth_csd_exec_dev')
    return n.value


def _csd_partial_dev(self, dx, dy, nsamples, sums_dptr):
    """Raw sums [sum|X|^2 | sum|Y|^2 | sum conj(X)Y re,im] (4 * nfft floats, natural order) of this time chunk."""
    n = C.c_uint64()
    self.ctx.check(self.ctx.lib.oth_csd_partial_dev(self.h, C.c_void_p(dx), C.c_void_p(dy), nsamples,
                                                    C.c_void_p(sums_dptr), C.byref(n)), 'oth_csd_partial_dev')
    return n.value


def _csd_scale_dev(self, sums_dptr, nseg_total, pxx=0, pyy=0, pxy=0, cxy=0):
    vp = lambda v: C.c_void_p(v) if v else None      # noqa: E731
    self.ctx.check(self.ctx.lib.oth_csd_scale_dev(self.h, C.c_void_p(sums_dptr), int(nseg_total), vp(pxx), vp(pyy),
                                                  vp(pxy), vp(cxy)), 'oth_csd_scale_dev')


WelchPlan.csd_exec_dev = _csd_exec_dev
WelchPlan.csd_partial_dev = _csd_partial_dev
WelchPlan.csd_scale_dev = _csd_scale_dev


class Chain(object):
    """stream_to_vector -> keep_one_in_n -> fft_vcc -> |.|/|.|^2 [-> IIR -> log] with GNU Radio's
    streaming state kept on the device (oth_chain)."""

    def __init__(self, ctx, nfft, window, fftshift, epilogue, keep_one_in_n):
        self.ctx = ctx
        self.nfft = int(nfft)
        w = None
        if window is not None and len(window):
            w = np.ascontiguousarray(window, np.float32)
            if w.shape != (self.nfft,):
                raise ValueError('window must have nfft entries')
        h = C.c_void_p()
        ctx.check(ctx.lib.oth_chain_create(ctx.h, self.nfft, _fptr(w) if w is not None else None,
                                           1 if fftshift else 0, int(epilogue), int(keep_one_in_n), C.byref(h)),
                  'oth_chain_create')
        self.h = h

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.oth_chain_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_keep_one_in_n(self, n):
        self.ctx.check(self.ctx.lib.oth_chain_set_keep_one_in_n(self.h, int(n)), 'oth_chain_set_keep_one_in_n')

    def set_iir_log(self, alpha, k_db):
        self.ctx.check(self.ctx.lib.oth_chain_set_iir_log(self.h, float(alpha), float(k_db)),
                       'oth_chain_set_iir_log')

    def set_kernel(self, which):
        self.ctx.check(self.ctx.lib.oth_chain_set_kernel(self.h, int(which)), 'oth_chain_set_kernel')

    def set_peak_hold(self, on):
        self.ctx.check(self.ctx.lib.oth_chain_set_peak_hold(self.h, 1 if on else 0), 'oth_chain_set_peak_hold')

    def reset(self):
        self.ctx.check(self.ctx.lib.oth_chain_reset(self.h), 'oth_chain_reset')

    def push(self, x, max_rows=None):
        """Feed samples; returns (rows, nrows_produced).  rows holds the LAST min(nrows, max_rows) rows."""
        x = _c64(x)
        cap = (len(x) // self.nfft + 2) if max_rows is None else int(max_rows)
        rows = np.empty((max(cap, 1), self.nfft), np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_push(self.h, x.ctypes.data_as(_p), len(x), 0, _fptr(rows), cap,
                                                   C.byref(n)), 'oth_chain_push')
        got = min(int(n.value), cap)
        return rows[:got], int(n.value)

    def push_dev(self, dptr, nsamples, rows_dptr=0, capacity=0):
        """Asynchronous: device-resident samples in, the last `capacity` rows to rows_dptr (device).  -> rows produced."""
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_push_dev(self.h, C.c_void_p(dptr), nsamples,
                                                       C.c_void_p(rows_dptr) if rows_dptr else None, int(capacity),
                                                       C.byref(n)), 'oth_chain_push_dev')
        return int(n.value)

    def push_async(self, x):
        """work() form: enqueue and return a ticket; the latest row is collected with poll() / wait()."""
        x = _c64(x)
        t = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_push_async(self.h, x.ctypes.data_as(_p), len(x), C.byref(t)),
                       'oth_chain_push_async')
        return int(t.value)

    def last_push_ops(self):
        """Stream operations (asynchronous copies + kernel launches) the last push enqueued."""
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_last_push_ops(self.h, C.byref(n)), 'oth_chain_last_push_ops')
        return int(n.value)

    def ticket_rows(self, ticket):
        """Rows the push behind `ticket` produces (known at enqueue time; never waits)."""
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_ticket_rows(self.h, int(ticket), C.byref(n)), 'oth_chain_ticket_rows')
        return int(n.value)

    def poll(self, ticket):
        """-> None while the GPU is still working, else (row or None, rows produced by that push)."""
        row = np.empty(self.nfft, np.float32)
        n, ready = C.c_uint64(), C.c_int()
        self.ctx.check(self.ctx.lib.oth_chain_poll(self.h, int(ticket), _fptr(row), C.byref(n), C.byref(ready)),
                       'oth_chain_poll')
        if not ready.value:
            return None
        return (row if n.value else None), int(n.value)

    def wait(self, ticket):
        row = np.empty(self.nfft, np.float32)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.oth_chain_wait(self.h, int(ticket), _fptr(row), C.byref(n)), 'oth_chain_wait')
        return (row if n.value else None), int(n.value)

    def peak(self):
        out = np.empty(self.nfft, np.float32)
        self.ctx.check(self.ctx.lib.oth_chain_get_peak(self.h, _fptr(out)), 'oth_chain_get_peak')
        return out

    def iir(self):
        out = np.empty(self.nfft, np.float32)
        self.ctx.check(self.ctx.lib.oth_chain_get_iir(self.h, _fptr(out)), 'oth_chain_get_iir')
        return out


_default_ctx = None


def default_context():
    """Process-wide context on device $OFDM_TOOLS_HIP_DEVICE (default 0, or LOCAL_RANK)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get('OFDM_TOOLS_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        _default_ctx = Context(dev)
    return _default_ctx
