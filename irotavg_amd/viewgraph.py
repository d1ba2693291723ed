"""Python mirror of the rotation side of the reference's ViewGraph / Pose API
(src/ViewGraph.hpp:54-75, src/Pose.hpp:35-59) over the C ABI (irotavg_viewgraph_*): same method
names and meaning (processFrame's vision front-end is out of scope -- views and connections are
fed in directly, cf. View::connect, src/ViewGraph.cpp:1438-1455)."""
import ctypes as C

import numpy as np

from . import capi


class ViewGraph:
    def __init__(self, **opts):
        self._h = C.c_void_p()
        o = capi.default_options(**opts)
        capi.check(capi.lib().irotavg_viewgraph_create(C.byref(self._h), C.byref(o)), "viewgraph_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().irotavg_viewgraph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _r(R):
        return np.ascontiguousarray(R, dtype=np.float64).reshape(9)

    def addView(self, R=None):
        """Appends a view (the rotation part of processFrame's `m_views.push_back`). Returns its id."""
        if R is None:
            rc = capi.lib().irotavg_viewgraph_add_view(self._h, None)
        else:
            r = self._r(R)
            rc = capi.lib().irotavg_viewgraph_add_view(self._h, capi._d(r))
        if rc < 0:
            raise capi.IrotavgError(rc, "add_view")
        return rc

    def numViews(self):
        return capi.lib().irotavg_viewgraph_num_views(self._h)

    def connect(self, i, j, Rij):
        """View::connect: R_j = R_ij R_i for i < j. True if added, False if already connected."""
        r = self._r(Rij)
        rc = capi.lib().irotavg_viewgraph_connect(self._h, int(i), int(j), capi._d(r))
        if rc < 0:
            raise capi.IrotavgError(rc, "connect")
        return bool(rc)

    def fixPose(self, idx, R):
        r = self._r(R)
        capi.check(capi.lib().irotavg_viewgraph_fix_pose(self._h, int(idx), capi._d(r)), "fixPose")

    def isPoseFixed(self, idx):
        rc = capi.lib().irotavg_viewgraph_is_pose_fixed(self._h, int(idx))
        if rc < 0:
            raise capi.IrotavgError(rc, "isPoseFixed")
        return bool(rc)

    def countFixedPoses(self):
        return capi.lib().irotavg_viewgraph_count_fixed_poses(self._h)

    def R(self, idx):
        r = np.zeros(9)
        capi.check(capi.lib().irotavg_viewgraph_get_pose(self._h, int(idx), capi._d(r)), "get_pose")
        return r.reshape(3, 3)

    def setR(self, idx, R):
        r = self._r(R)
        capi.check(capi.lib().irotavg_viewgraph_set_pose(self._h, int(idx), capi._d(r)), "set_pose")

    def rotAvg(self, winSize):
        """ViewGraph::rotAvg (src/ViewGraph.cpp:1263-1435). Returns the info record as a dict."""
        info = capi.RotAvgInfo()
        capi.check(capi.lib().irotavg_viewgraph_rot_avg(self._h, int(winSize), C.byref(info)), "rotAvg")
        return {k: getattr(info, k) for k, _ in capi.RotAvgInfo._fields_}

    def prepare(self):
        """irotavg_viewgraph_prepare: a dry run of the global re-solve (no pose changes) that takes the one-time
        costs of a process out of the first loop closure's latency."""
        capi.check(capi.lib().irotavg_viewgraph_prepare(self._h), "prepare")

    def savePoses(self, filename, t=None):
        """ViewGraph::savePoses (src/ViewGraph.cpp:1206-1231); t: optional (n, 3) translations."""
        tp = None
        if t is not None:
            t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
            tp = capi._d(t)
        capi.check(capi.lib().irotavg_viewgraph_save_poses(self._h, str(filename).encode(), tp), "savePoses")


def rotAvgBatch(graphs, winSize):
    """rotAvg for several DIFFERENT view-graphs at once (irotavg_viewgraph_rot_avg_batch): the windows that fit the
    wave-resident kernel are solved by one launch, a workgroup per window. Returns one info dict per graph."""
    n = len(graphs)
    hs = (C.c_void_p * n)(*[g._h for g in graphs])
    infos = (capi.RotAvgInfo * n)()
    capi.check(capi.lib().irotavg_viewgraph_rot_avg_batch(hs, n, int(winSize), infos), "rotAvgBatch")
    return [{k: getattr(infos[b], k) for k, _ in capi.RotAvgInfo._fields_} for b in range(n)]


def rmat2quat(R):
    """src/ViewGraph.cpp:1175-1203; row-major 3x3 -> [x y z w]."""
    r = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    q = np.zeros(4)
    capi.lib().irotavg_rmat2quat(capi._d(r), capi._d(q))
    return q


def quat2rmat(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    r = np.zeros(9)
    capi.lib().irotavg_quat2rmat(capi._d(q), capi._d(r))
    return r.reshape(3, 3)
