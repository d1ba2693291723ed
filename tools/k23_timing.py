import sys, numpy as np
sys.path.insert(0,'.')
import bench
from irotavg_amd import capi
S,Q0=bench.build_problem(100000,2000000,0.0,0)
G=capi.Graph(S["I"],S["QQ"],S["n"],1); G.set_rotations(Q0); G.irls(4,bench.SIG,100,1e-3)
for w,name,by in ((1,'K1',131.2e6),(2,'K2',82.4e6),(3,'K3',115.2e6)):
    ms=min(G.time_kernel(w,50) for _ in range(3)); print(name,'%.2f us %.0f GB/s frac %.3f'%(ms*1e3,by/ms/1e6,by/ms/1e6/8000))
