import sys, time, numpy as np
sys.path.insert(0,'.')
from irotavg_amd import capi, synth
from oracle import oracle as O
SIG=5*np.pi/180
n=100000
for deg in (4, 5, 8, 10, 15, 20, 30):
    S = synth.make_graph(n, n*deg, 0.0, seed=0)
    Q = np.zeros((n,4)); Q[:,3]=1; Q[0]=S["Qgt"][0]
    rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
    out=[]
    for om,kc in ((0.7,1.0),(0.7,1.3),(0.7,1.6),(0.7,2.0),(0.7,2.4)):
        with capi.Graph(S["I"], S["QQ"], n, 1, mg_omega=om, mg_kc=kc) as G:
            best=1e9
            for rep in range(2):
                G.set_rotations(Qm); G.reset_stats()
                t=time.perf_counter(); r=G.irls(4,SIG,100,1e-3); dt=time.perf_counter()-t
                best=min(best,dt)
            st=G.stats()
            out.append("kc %.1f: %.1f it %.2f ms"%(kc,st["pcg_iters"]/st["pcg_solves"],best*1e3))
    print("band deg %2d levels %s "%(deg, st["level_rows"][:st["levels"]])+" | ".join(out), flush=True)
# the 10k case: does kc matter?
S = synth.make_graph(10000, 150000, 0.0, seed=0)
Q = np.zeros((10000,4)); Q[:,3]=1; Q[0]=S["Qgt"][0]
rc, Qm = O.init_mst(Q, S["QQ"], S["I"], 1)
for kc in (0.5, 1.0, 3.0):
    with capi.Graph(S["I"], S["QQ"], 10000, 1, mg_kc=kc) as G:
        G.set_rotations(Qm); r=G.irls(4,SIG,100,1e-3); st=G.stats()
        print("10k kc", kc, "pcg iters", st["pcg_iters"], "solves", st["pcg_solves"], "levels", st["level_rows"][:st["levels"]])
