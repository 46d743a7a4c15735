"""Two-channel sums (oth_csd_exec) at 32768 / 65536 points, 2 x 2^26 device-resident samples: the register radix-16 route
against the coverage kernels (tuning "anycov").  gpurun -- python tools/csd_big.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, 'gr-ofdm_tools_amd')
from ofdm_tools import _hip
ctx = _hip.Context()
for n in (32768, 65536):
    for var in (None, 'anycov'):
        ns = 1 << 26
        g = torch.Generator(device='cuda').manual_seed(1)
        x = torch.randn(ns, 2, device='cuda', generator=g); y = torch.randn(ns, 2, device='cuda', generator=g)
        plan = ctx.welch_plan(n)
        if var: plan.set_tuning(var)
        import ctypes
        f = plan.csd_device_src
        t = []
        for i in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            f(x.data_ptr(), y.data_ptr(), ns)
            torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        ms = min(t) * 1e3
        print(n, var, plan.last_recipe().split()[0], '%.3f ms' % ms, '%.1f %% of 8 TB/s' % (2 * ns * 8 / (ms * 1e-3) / 8e12 * 100), flush=True)
