"""K1 (k_edge_residual) at 100k / 2M with the Infinity Cache cold: a 1 GB fill between the launches. Run under
`rocprofv3 --kernel-trace --stats`: the kernel's average is the cold-cache time, to be read next to the back-to-back time
(working set inside the 256 MiB cache) and the in-situ time inside irls (behind a solve that moved 160 MB).
FETCH_SIZE cannot tell the cases apart: Infinity-Cache hits are counted as fetches (MI355X_MICROARCH.md, HBM section)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from irotavg_amd import capi, ral, synth
n, m = 100000, 2000000
S = synth.make_graph(n, m, 0.0, seed=0)
Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
ral.init_mst(Q0, S["QQ"], S["I"], 1)
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
with capi.Graph(S["I"], S["QQ"], n, 1) as G:
    G.set_rotations(Q0)
    for mode in ("cold", "warm"):
        for _ in range(30):
            if mode == "cold":
                big.fill_(1.0)
                torch.cuda.synchronize()
            G.edge_residual()
            G.synchronize()
    print("done: 30 cold launches (each behind a 1 GiB fill) then 30 warm ones; read k_edge_residual in the trace, "
          "first 30 dispatches = cold")
