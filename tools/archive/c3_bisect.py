#!/usr/bin/env python3
"""Why does csd4096ws_kernel take 0.336-0.352 ms under bench.py::csd_bench and 0.310-0.316 ms under
tools/prof_driver.py C3 (round-4 verdict, weak 3)?  One process, one GPU: the same 2 x 2^26 launch under every
combination of the differences between the two harnesses, kernel time by the library's own HIP events
(oth_ctx_set_timing: the events bracket the transform launch only).

  stream   own   = context on its own non-blocking stream, hipMalloc'ed buffers    (prof_driver)
           torch = context on torch's current stream, torch.empty buffers          (bench.py)
  data     indep = y from its own seed (prof_driver)      roll = y = 0.7 roll(x, 5) + 0.5 noise (bench.py)
  call     host  = oth_csd_exec(src_is_device=1): host outputs, a stream synchronisation per call   (prof_driver)
           dev   = oth_csd_exec_dev: device outputs, launches back to back                            (bench.py)
  reps     3 (prof_driver's default) or 50 (bench.py: 200 // 4)
  soak     0 / 1: half a second of the 2^28-sample C2 launch right before (bench.py runs C2 and the sweep first)
"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import torch  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
DC = 0.1 + 0.05j
n = 1 << 26
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
tstream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(tstream)
hann = windows.get_window('hann', 4096)
ctxs = {'own': _hip.Context(0), 'torch': _hip.Context(0, stream=tstream.cuda_stream)}


def _loaded_hip():
    import ctypes as C
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError('no HIP runtime mapped')


def d2d(dst, src, nbytes):
    import ctypes as C
    hip = _loaded_hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), nbytes, 3)      # hipMemcpyDeviceToDevice
    if rc:
        raise RuntimeError('hipMemcpy %d' % rc)


def buffers(kind, data):
    ctx = ctxs[kind]
    if kind == 'own':
        dx, dy = ctx.alloc(n * 8), ctx.alloc(n * 8)
        keep = None
    else:
        x = torch.empty((n, 2), dtype=torch.float32, device=dev)
        y = torch.empty((n, 2), dtype=torch.float32, device=dev)
        dx, dy, keep = x.data_ptr(), y.data_ptr(), (x, y)
    ctx.synth_iq(dx, n, 1003, TONES, DC)
    if data == 'indep':
        ctx.synth_iq(dy, n, 1004, TONES, DC)
    else:
        ctx.synth_iq(dy, n, 1004, (), 0j)
        ctx.sync()
        torch.cuda.synchronize()
        # y = 0.7 roll(x, 5) + 0.5 noise, formed by torch on views of whatever memory holds the samples
        if keep is None:
            # the mix is formed by torch on copies and written back into the hipMalloc'ed buffer (device-to-device copies
            # through the HIP runtime this process already has loaded)
            tx = torch.empty((n, 2), dtype=torch.float32, device=dev)
            ty = torch.empty((n, 2), dtype=torch.float32, device=dev)
            d2d(tx.data_ptr(), dx, n * 8)
            d2d(ty.data_ptr(), dy, n * 8)
            ty.mul_(0.5).add_(torch.roll(tx, 5, 0), alpha=0.7)
            torch.cuda.synchronize()
            d2d(dy, ty.data_ptr(), n * 8)
            del tx, ty
            torch.cuda.empty_cache()
        else:
            keep[1].mul_(0.5).add_(torch.roll(keep[0], 5, 0), alpha=0.7)
            torch.cuda.synchronize()
    return dx, dy, keep


def soak(ctx, secs=0.5):
    m = 1 << 28
    d, o = ctx.alloc(m * 8), ctx.alloc(4096 * 4)
    ctx.synth_iq(d, m, 1002, TONES, DC)
    plan = ctx.welch_plan(4096, window=hann, fs=1.0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(8):
            plan.exec_dev(d, m, o)
        ctx.sync()
    plan.close()
    ctx.free(d)
    ctx.free(o)


def measure(kind, data, call, reps, do_soak, bufs):
    ctx = ctxs[kind]
    dx, dy, _ = bufs
    plan = ctx.welch_plan(4096, window=hann, fs=1.0)
    outs = [ctx.alloc(4096 * 4 * (2 if i == 2 else 1)) for i in range(4)]
    if call == 'host':
        run = lambda: plan.csd_device_src(dx, dy, n)      # noqa: E731
    else:
        run = lambda: plan.csd_exec_dev(dx, dy, n, *outs)      # noqa: E731
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(4):
            run()
        ctx.sync()
    if do_soak:
        soak(ctx)
    ctx.set_timing(True)
    ctx.get_timing()
    for _ in range(reps):
        run()
    ms, k = ctx.get_timing()
    ctx.set_timing(False)
    for o in outs:
        ctx.free(o)
    plan.close()
    return ms / max(k, 1)


print('stream data  call reps soak   kernel ms   %% of 8 TB/s (16 B/pair)')
for kind, data in itertools.product(('own', 'torch'), ('indep', 'roll')):
    bufs = buffers(kind, data)
    for call, reps, sk in (('host', 3, 0), ('host', 50, 0), ('dev', 3, 0), ('dev', 50, 0), ('dev', 50, 1), ('host', 3, 1)):
        ms = measure(kind, data, call, reps, sk, bufs)
        print('%-6s %-5s %-4s %4d %4d   %.4f      %.1f' % (kind, data, call, reps, sk, ms, 16.0 * n / ms / 1e6 / 80.0), flush=True)
    if bufs[2] is None:
        ctxs[kind].free(bufs[0])
        ctxs[kind].free(bufs[1])
    del bufs
    torch.cuda.empty_cache()
