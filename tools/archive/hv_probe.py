#!/usr/bin/env python3
"""Where do the ~40-50 us between bench.py's `ms_per_step` and `host_visible_ms_per_step` go?  One blocking call per step on
the 2^28-sample C2 launch, median host-clock time, right after a sustained soak (as in bench.py), for: context stream (own /
torch's) x completion (host row + polled word | device row + hipStreamSynchronize | device row + torch.cuda.synchronize)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import torch  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
n = 1 << 28
iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
out = torch.zeros(4096, dtype=torch.float32, device=dev)
hann = windows.get_window('hann', 4096)
for name, ctx in (('own stream', _hip.Context(0)), ('torch stream', _hip.Context(0, stream=ts.cuda_stream))):
    ctx.synth_iq(iq.data_ptr(), n, 1002, ((0.5, 0.1234),), 0.1 + 0.05j)
    plan = ctx.welch_plan(4096, window=hann, fs=1.0)
    forms = {
        'host row + polled word': lambda: plan.exec_device_src(iq.data_ptr(), n),
        'device row + ctx.sync': lambda: (plan.exec_dev(iq.data_ptr(), n, out.data_ptr()), ctx.sync()),
        'device row + torch.cuda.synchronize': lambda: (plan.exec_dev(iq.data_ptr(), n, out.data_ptr()), torch.cuda.synchronize()),
    }
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:                      # soak: back to back, as bench.py's timed steps
        for _ in range(8):
            plan.exec_dev(iq.data_ptr(), n, out.data_ptr())
        ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        plan.exec_dev(iq.data_ptr(), n, out.data_ptr())
    ctx.sync()
    pipe = (time.perf_counter() - t0) * 10.0
    res = {k: [] for k in forms}
    for _ in range(60):                                         # interleaved
        for k, f in forms.items():
            t0 = time.perf_counter()
            f()
            res[k].append((time.perf_counter() - t0) * 1e3)
    print('%-13s pipelined %.4f ms | ' % (name, pipe) + ' | '.join('%s %.4f (min %.4f)' % (k, statistics.median(v), min(v)) for k, v in res.items()), flush=True)
    plan.close()
