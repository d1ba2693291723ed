"""Development aid: where a launch of k_bcr_reduce spends its time -- shader-clock stamps of thread 0 of one chunk at the
phase boundaries (bcr_stamp in bcr.hip; irotavg_graph_time_kernel 200 + 16 level + slot), next to the launch times."""
import sys
sys.path.insert(0, ".")
import numpy as np
from irotavg_amd import capi, synth, ral

NAMES = ["start", "zeroed", "loaded", "r0 sweep", "r0 W,right", "r0 left mul", "r0 left put", "r1 sweep", "r1 W,right",
         "r1 left mul", "r1 left put", "r2 sweep", "r2 W,right", "r2 left mul", "r2 left put", "stored"]
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 2000000)
S = synth.make_graph(n, m, 0.0, seed=0)
Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
ral.init_mst(Q, S["QQ"], S["I"], 1)
with capi.Graph(S["I"], S["QQ"], n, 1) as G:
    G.set_rotations(Q)
    r = G.irls(4, 5 * np.pi / 180, 50, 1e-3)
    info = G.direct_info()
    nl = len(info["levels"])
    s = min(G.time_kernel(19, 30) for _ in range(3))
    red = [min(G.time_kernel(20 + l, 50) for _ in range(3)) for l in range(nl)]
    back = [min(G.time_kernel(40 + l, 50) for _ in range(3)) for l in range(nl)]
    print(n, m, "block", info["block"], "levels", [(L["blocks"], L["chunks"]) for L in info["levels"]], "iters", r["iters"],
          "score %.9e" % r["scores"][-1])
    print("solve %.1f us; reduce %s; back %s" % (1e3 * s, ["%.1f" % (1e3 * b) for b in red], ["%.1f" % (1e3 * b) for b in back]))
    for l in range(nl):
        st = [G.time_kernel(200 + 16 * l + k, 1) for k in range(16)]
        tot = st[15]
        prev = 0.0
        parts = []
        for k in range(1, 16):
            if st[k] < 0:
                continue
            parts.append("%s %+d" % (NAMES[k], st[k] - prev))
            prev = st[k]
        print("level %d chunk 0: %d clocks in the kernel (launch %.1f us): %s" % (l, tot, 1e3 * red[l], "; ".join(parts)))

    # the single-launch upper reduction (k_bcr_reduce_up): workgroup 0's stamps per level of the launch
    UP = {16: "entered", 17: "waited", 18: "body done", 19: "counted"}
    try:
        for i in range(nl - 1):
            st = [G.time_kernel(600 + 32 * i + k, 1) for k in range(28)]
            print("   raw HW_ID:", ["%x" % (int(st[20 + w]) - 0x10000) if st[20 + w] >= 0 else None for w in range(8)])
            print("   waves' (SIMD, slot) of workgroup 0:", [((int(st[20 + w]) >> 4) & 3, int(st[20 + w]) & 15) if st[20 + w] >= 0 else None for w in range(8)])
            body = ["%d:%+d" % (k, st[k] - st[k - 1]) for k in range(1, 16) if st[k] >= 0 and st[k - 1] >= 0]
            print("fused up, level %d of the launch: entered at %d, waited until %d, body done %s, counted %s; body stamps (slot:delta) %s"
                  % (i + 1, st[16], st[17], st[18], st[19], " ".join(body)))
    except Exception as e:  # not a handle that uses the single launch
        print("no fused upper reduction:", e)
