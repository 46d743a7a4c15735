"""welch32k.hip against scipy's form in f64 (tests/test_anylen_gpu.py holds the gated version): quick look at both lengths,
then the time per call on 2^27 resident samples against the four-step route (tuning "r16")."""
import sys, time, numpy as np, torch
sys.path.insert(0, 'gr-ofdm_tools_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from ofdm_tools import _hip
from oracle import ref_cpu as R
import scipy.signal as sg
ctx = _hip.Context()
def relerr(a, b): return float(np.max(np.abs(a - b) / b))
for n in (65536, 32768):
    for detrend in (True, False):
        for nseg, ov in ((1, n // 2), (3, n // 2), (9, n // 2), (300, n // 2), (7, 1000), (5, 0)):
            step = n - ov
            x = R.synth_iq(n + step * (nseg - 1) + 17, 5, dc=3 + 2j)
            w = sg.get_window('hann', n).astype(np.float32)
            plan = ctx.welch_plan(n, nperseg=n, noverlap=ov, window=w, detrend=detrend)
            got = plan.exec(x)
            _, ref = R.welch_np(x, nperseg=n, nfft=n, noverlap=ov, detrend='constant' if detrend else False)
            print(n, detrend, nseg, ov, plan.last_recipe().split()[0], plan.last_nseg, 'relerr %.2e' % relerr(got, ref), flush=True)
            plan.close()
ns = 1 << 27
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(ns, 2, device='cuda', generator=g)
for n in (65536, 32768):
    for var in (None, 'r16'):
        plan = ctx.welch_plan(n)
        if var: plan.set_tuning(var)
        out = torch.zeros(n, dtype=torch.float32, device='cuda')
        for i in range(30): plan.exec_dev(x.data_ptr(), ns, out.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): plan.exec_dev(x.data_ptr(), ns, out.data_ptr())
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print(n, var, plan.last_recipe().split()[0], '%.3f ms' % ms, '%.1f %% of 8 TB/s' % (ns * 8 / (ms * 1e-3) / 8e12 * 100), flush=True)
