"""welch32k.hip timing probes: the C2-like stream (2^27 samples, 50 % overlap: 8191 segments) against the same number of
segments cut from a 320 KiB buffer (step 1: every load is an L2 hit - what the kernel costs without HBM)."""
import sys, time, numpy as np, torch
sys.path.insert(0, 'gr-ofdm_tools_amd')
from ofdm_tools import _hip
ctx = _hip.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(1 << 27, 2, device='cuda', generator=g)
def run(label, ns, **kw):
    plan = ctx.welch_plan(n, **kw)
    var = kw.get('var')
    t = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.exec_device_src(x.data_ptr(), ns)
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    ms = min(t) * 1e3
    print('%-28s %s nseg %d  %.3f ms  %.2f us per segment and CU' % (label, plan.last_recipe().split()[0], plan.last_nseg, ms, ms * 1e3 / (plan.last_nseg / 256.0)), flush=True)
    plan.close()
run('stream, 50 % overlap', 1 << 27)
run('stream, no overlap', 1 << 27, noverlap=0)
run('L2-resident (step 1)', n + 8190, noverlap=n - 1)
run('stream, 50 %, no detrend', 1 << 27, detrend=0)
run('L2-resident, no detrend', n + 8190, noverlap=n - 1, detrend=0)
