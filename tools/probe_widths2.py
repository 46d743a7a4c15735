import sys
sys.path.insert(0,'gr-ofdm_tools_amd')
from ofdm_tools import _hip
ctx=_hip.Context(0)
n=1<<28
d=ctx.alloc(n*8)
ctx.synth_iq(d,n,1,(),0j)
for rep in (5,-5,5,-5):
    ms=ctx.stream_read_probe(d,n*8,rep)
    print('repeats',rep,'%.1f GB/s'%(n*8/ms/1e6))
