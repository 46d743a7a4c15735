"""Same-box A/B of the one-workgroup route (default; "td" forces the time-domain detrend, which is all this route has) against
the four-step ("r16"): interleaved launches on 2^27 resident samples."""
import sys, time, numpy as np, torch
sys.path.insert(0, 'gr-ofdm_tools_amd')
from ofdm_tools import _hip
ctx = _hip.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ns = 1 << 27
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(ns, 2, device='cuda', generator=g)
out = torch.zeros(n, dtype=torch.float32, device='cuda')
plans = {}
for var in (None, 'td', 'r16'):
    p = ctx.welch_plan(n)
    if var: p.set_tuning(var)
    plans[var] = p
for rep in range(3):
    for var, p in plans.items():
        for i in range(20): p.exec_dev(x.data_ptr(), ns, out.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): p.exec_dev(x.data_ptr(), ns, out.data_ptr())
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print(rep, var, p.last_recipe().split()[0], p.last_recipe().split()[2], '%.4f ms' % ms, '%.1f %%' % (ns * 8 / (ms * 1e-3) / 8e12 * 100), flush=True)
