import sys,time,statistics
sys.path.insert(0,"gr-ofdm_tools_amd")
from ofdm_tools import _hip, windows
ctx=_hip.Context(0); n=1<<28; d=ctx.alloc(n*8); ctx.synth_iq(d,n,1002,(),0j)
plan=ctx.welch_plan(4096, window=windows.get_window("hann",4096), fs=1.0)
for _ in range(10): plan.exec_device_src(d,n)
t=[]
for _ in range(60):
    t0=time.perf_counter(); plan.exec_device_src(d,n); t.append((time.perf_counter()-t0)*1e3)
n2=1<<20
t2=[]
for _ in range(200):
    t0=time.perf_counter(); plan.exec_device_src(d,n2); t2.append((time.perf_counter()-t0)*1e3)
print("host visible ms: 2^28 %.4f   2^20 %.4f" % (statistics.median(t), statistics.median(t2)))
