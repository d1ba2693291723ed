"""CPU ORACLE (test infrastructure) for the caller of the hot path: a literal restatement of
ViewGraph::rotAvg (src/ViewGraph.cpp:1263-1435) and of fixPose / isPoseFixed / countFixedPoses
(:1234-1260) on top of the C oracle (oracle.py). PARITY UNPINNED (see irotavg_oracle.h).

Connections of a view are walked by ascending neighbour id (the reference walks a
std::map<View*,...> in pointer order, which is not reproducible; src/View.hpp:66)."""
import numpy as np

from . import oracle as O


class ViewGraphOracle:
    def __init__(self):
        self.R = []          # absolute rotations, 3x3 row-major (Pose::R)
        self.fixed = []      # m_fixed_mask
        self.conn = []       # per view: {neighbour id: R_ij of the pair (min, max)}

    def addView(self, R=None):
        self.R.append(np.eye(3) if R is None else np.array(R, dtype=np.float64).reshape(3, 3))
        self.fixed.append(False)
        self.conn.append({})
        return len(self.R) - 1

    def connect(self, a, b, Rij):  # View::connect :1438-1455
        if b in self.conn[a]:
            return False
        Rij = np.array(Rij, dtype=np.float64).reshape(3, 3)
        self.conn[a][b] = Rij
        self.conn[b][a] = Rij
        return True

    def fixPose(self, idx, R):  # :1234-1245
        self.fixed[idx] = True
        self.R[idx] = np.array(R, dtype=np.float64).reshape(3, 3)

    def isPoseFixed(self, idx):
        return self.fixed[idx]

    def countFixedPoses(self):
        return sum(self.fixed)

    def rotAvg(self, winSize):  # :1263-1435
        assert winSize > 2
        m = len(self.R)
        win = min(m, winSize)
        if win < 2:
            return dict(skipped=1)
        I, qq, vertices = [], [], set()
        for t in range(m - win, m):
            j = t
            for i in sorted(self.conn[j]):
                if i < j:
                    I.append((i, j))
                    vertices.update((i, j))
                    qq.append(O.rmat2quat(self.conn[j][i]))
        ne, nv = len(qq), len(vertices)
        if ne < win:
            return dict(skipped=2)
        if nv < win:
            return dict(skipped=3)
        f = nv - win
        for x in sorted(vertices):
            if x >= m - win and self.fixed[x]:
                f += 1
        v2i, i2v = {}, {}
        t, k = 0, f
        for x in sorted(vertices):
            if x >= m - win and not self.fixed[x]:
                i2v[k] = x; v2i[x] = k; k += 1
            else:
                i2v[t] = x; v2i[x] = t; t += 1
        I = np.array([(v2i[a], v2i[b]) for a, b in I], dtype=np.int32)
        Q = np.zeros((nv, 4))
        for x in vertices:
            Q[v2i[x]] = O.rmat2quat(self.R[x])
        if f == 0:
            Q[0] = [0, 0, 0, 1]
            f = 1
        if nv - f < 1:
            return dict(skipped=4)
        QQ = np.array(qq)
        a = O.l1ra(QQ, I, Q, f, 100, .001)
        b = O.irls(QQ, I, a["Q"], f, 4, 5 * np.pi / 180, 100, .001)
        for r in range(f, nv):
            self.R[i2v[r]] = O.quat2rmat(b["Q"][r])
        return dict(skipped=0, n_views=nv, n_edges=ne, n_fixed=f, l1_iters=a["iters"],
                    irls_iters=b["iters"])
