"""Development aid: where the workgroups of k_bcr_reduce_up (the single-launch reduction of all levels above level 0) spend
their time -- shader clocks of workgroup IROTAVG_BCR_STAMP_CHUNK (default 0) per level of the launch, relative to the moment
it entered the launch's first level (irotavg_graph_time_kernel 600 + 32 level + slot, bcr_stamps_up in bcr.hip)."""
import os
import sys
sys.path.insert(0, ".")
import numpy as np
from irotavg_amd import capi, synth, ral

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 2000000)
wgs = [int(a) for a in sys.argv[3:]] or [0]
S = synth.make_graph(n, m, 0.0, seed=0)
Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
ral.init_mst(Q, S["QQ"], S["I"], 1)
with capi.Graph(S["I"], S["QQ"], n, 1) as G:
    G.set_rotations(Q)
    G.irls(4, 5 * np.pi / 180, 50, 1e-3)
    info = G.direct_info()
    print(n, m, "block", info["block"], "levels", [(L["blocks"], L["chunks"]) for L in info["levels"]])
    nl = len(info["levels"])
    for wg in wgs:
        os.environ["IROTAVG_BCR_STAMP_CHUNK"] = str(wg)
        for i in range(nl - 1):
            if wg >= info["levels"][1 + i]["chunks"]:
                break
            st = [G.time_kernel(600 + 32 * i + k, 1) for k in range(20)]
            body = ["%d:%+d" % (k, st[k] - st[k - 1]) for k in range(1, 16) if st[k] >= 0 and st[k - 1] >= 0]
            print("workgroup %d, level %d of the solve: entered %d, wait over %d, body done %s, counted %s; body (slot:clocks since the slot before) %s"
                  % (wg, i + 1, st[16], st[17], st[18], st[19], " ".join(body)))
