"""The host-side alternatives of round 3 give the same numbers: read-backs by stream synchronise instead of the polled
pinned block (IROTAVG_NO_POLL, read once per process), the upload of a one-shot call on the calling thread instead of
helper threads (IROTAVG_UPLOAD_THREADS=0), pinned staging without the mapped / coherent flags."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from irotavg_amd import capi, ral, synth
n, m = 12000, 120000
S = synth.make_graph(n, m, 0.0, seed=4)
Q = np.zeros((n, 4)); Q[:, 3] = 1; Q[0] = S["Qgt"][0]
ral.init_mst(Q, S["QQ"], S["I"], 1)
with capi.Graph(S["I"], S["QQ"], n, 1) as G:
    G.set_rotations(Q)
    a = G.l1ra(2, 1e-3)
    b = G.irls(4, 5 * np.pi / 180, 30, 1e-3)
    Qh = G.get_rotations()
# the one-shot call (handle built on the device, upload inside)
Q1, w1 = Q.copy(), np.zeros(m)
it1, _ = ral.irls(S["QQ"], S["I"], None, 4, 5 * np.pi / 180, Q1, 1, 30, 1e-3, w1)
print(json.dumps({"l1": a["iters"], "irls": b["iters"], "scores": list(map(float, a["scores"])) + list(map(float, b["scores"])),
                  "q": Qh.ravel().tolist()[:4000], "one_shot_iters": int(it1), "q1": Q1.ravel().tolist()[:4000]}))
""" % ROOT


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_polled_and_synchronised_read_backs_and_upload_threads_agree():
    ref = run({})
    for extra in ({"IROTAVG_NO_POLL": "1"}, {"IROTAVG_UPLOAD_THREADS": "0"},
                  {"IROTAVG_NO_POLL": "1", "IROTAVG_PIN_DEFAULT": "1"}):
        got = run(extra)
        assert (got["l1"], got["irls"], got["one_shot_iters"]) == (ref["l1"], ref["irls"], ref["one_shot_iters"]), extra
        np.testing.assert_array_equal(got["scores"], ref["scores"])     # the same kernels, the same sums
        np.testing.assert_array_equal(got["q"], ref["q"])
        np.testing.assert_array_equal(got["q1"], ref["q1"])


def test_one_shot_calls_keep_the_handle_of_the_same_graph_and_notice_a_change_in_place():
    """The reference's callers pass the same (I, QQ) to l1ra and then to irls (src/ViewGraph.cpp:1400-1417,
    ral/test.cpp:295-301): the second one-shot call takes the handle the first one left (a hit, nothing uploaded but Q)
    and returns what two calls on fresh handles return, bit for bit. ONE entry of QQ changed in place between two calls
    is another graph: no hit, and the result is that of the changed graph."""
    import ctypes as C
    import numpy as np
    from irotavg_amd import capi, synth
    from oracle import oracle as O
    L = capi.lib()
    n, m = 6000, 90000
    S = synth.make_graph(n, m, 0.0, seed=5, p_band_out=0.02)
    Q0 = np.zeros((n, 4)); Q0[:, 3] = 1; Q0[0] = S["Qgt"][0]
    rc, Q0 = O.init_mst(Q0, S["QQ"], S["I"], 1)
    assert rc == 0
    Ie, QQf = capi.edges(S["I"]), capi.fmat(S["QQ"])     # the caller's arrays: the same two buffers in every call

    def pipeline(Qstart, QQarr):
        Qf = capi.fmat(Qstart.copy())
        it, rt = C.c_int(0), C.c_double(0)
        capi.check(L.irotavg_l1ra(m, n, 1, capi._i(Ie), capi._d(QQarr), m, capi._d(Qf), n, 3, 1e-3, C.byref(it),
                                  C.byref(rt)), "l1ra")
        w = np.zeros(m)
        it2 = C.c_int(0)
        capi.check(L.irotavg_irls(m, n, 1, capi._i(Ie), capi._d(QQarr), m, 4, 5 * np.pi / 180, capi._d(Qf), n, 50, 1e-3,
                                  capi._d(w), C.byref(it2), C.byref(rt)), "irls")
        return it.value, it2.value, np.array(Qf), w

    capi.oneshot_cache(False)
    ref = pipeline(Q0, QQf)
    capi.oneshot_cache(True)
    capi.oneshot_cache_clear()
    h0, m0 = capi.oneshot_cache_stats()
    out = pipeline(Q0, QQf)
    h1, m1 = capi.oneshot_cache_stats()
    assert (h1 - h0, m1 - m0) == (1, 1), (h0, m0, h1, m1)       # l1ra builds, irls takes the kept handle
    assert out[:2] == ref[:2]
    np.testing.assert_array_equal(out[2], ref[2])
    np.testing.assert_array_equal(out[3], ref[3])
    out2 = pipeline(Q0, QQf)                                       # the same graph again: two hits
    h2, m2 = capi.oneshot_cache_stats()
    assert (h2 - h1, m2 - m1) == (2, 0)
    np.testing.assert_array_equal(out2[2], ref[2])
    # one relative rotation replaced in place (same buffer, same pointer): a miss, and the changed graph's result
    k = m // 2
    QQchg = QQf            # (no copy)
    old = QQchg[k].copy()
    R = np.array([0.5, -0.5, 0.5, 0.5])
    QQchg[k] = R
    out3 = pipeline(Q0, QQchg)
    h3, m3 = capi.oneshot_cache_stats()
    assert (h3 - h2, m3 - m2) == (1, 1), (h2, m2, h3, m3)         # l1ra misses and rebuilds, irls hits the rebuilt one
    capi.oneshot_cache(False)
    ref3 = pipeline(Q0, QQchg)
    capi.oneshot_cache(True)
    np.testing.assert_array_equal(out3[2], ref3[2])
    np.testing.assert_array_equal(out3[3], ref3[3])
    assert np.abs(out3[3] - ref[3]).max() > 0                      # (and it IS another result)
    QQchg[k] = old
    capi.oneshot_cache_clear()
