import sys, time, numpy as np
sys.path.insert(0,'.')
from irotavg_amd import capi, synth
from oracle import oracle as O
SIG=5*np.pi/180
S = synth.make_graph(100000, 2000000, 0.02, seed=0)
Q = np.zeros((100000,4)); Q[:,3]=1; Q[0]=S["Qgt"][0]
rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
with capi.Graph(S["I"], S["QQ"], 100000, 1) as G:
    G.set_rotations(Qm); r=G.irls(4,SIG,100,1e-3)
    print("dense inversion ms:", G.time_kernel(7, 10))
    best=1e9
    for rep in range(4):
        G.set_rotations(Qm); G.reset_stats()
        t=time.perf_counter(); r=G.irls(4,SIG,100,1e-3); dt=time.perf_counter()-t
        best=min(best,dt)
    st=G.stats()
    print("iters=%d pcg/solve=%.1f  %.2f ms  %.0f M edge-upd/s"%(r["iters"],st["pcg_iters"]/st["pcg_solves"],best*1e3,2e6*r["iters"]/best/1e6), flush=True)
    Qa=G.get_rotations()
np.save("gpurun_out/_gj_Q_%s.npy"%sys.argv[1], Qa)
