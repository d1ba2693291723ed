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
