"""C-ABI library: loads without a GPU, exports every symbol include/irotavg_hip.h declares, and
the host-side entry points (init_mst, make_A) match the oracle. No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from irotavg_amd import capi, ral, synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "irotavg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(irotavg_[a-zA-Z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for s in names:
        assert hasattr(L, s), "missing export: " + s
    assert sorted(capi.SYMBOLS) == names          # the Python binding lists the same set
    assert b"gfx950" in L.irotavg_version()


def test_error_strings():
    L = capi.lib()
    assert L.irotavg_error_string(0) == b"ok"
    assert b"CPU fallback" in L.irotavg_error_string(capi.ERR_NO_DEVICE)
    assert b"span" in L.irotavg_error_string(capi.ERR_NOT_SPANNING)


def test_option_and_stats_struct_sizes_match_header():
    o = capi.default_options()
    assert (o.pcg_rtol, o.pcg_max_iters, o.mg_agg, o.mg_dense_max, o.device) == (1e-10, 2000, 0, 0, -1)
    assert o.mg_omega == pytest.approx(0.7) and o.mg_kc == 0.0
    assert o.band_direct == 0                    # a banded operator above 2048 free views is solved directly
    assert C.sizeof(capi.Options) == 88 and C.sizeof(capi.RotAvgInfo) == 40


def test_init_mst_matches_oracle_and_reports_non_spanning(fixture_graph):
    g = fixture_graph
    Q = g["Q"].copy()
    ral.init_mst(Q, g["QQ"], g["I"], 1)
    rc, Qo = O.init_mst(g["Q"], g["QQ"], g["I"], 1)
    np.testing.assert_array_equal(Q, Qo)
    I = np.array([[0, 1], [2, 3]], dtype=np.int32)
    QQ = np.tile([0, 0, 0, 1.0], (2, 1))
    with pytest.raises(capi.IrotavgError) as e:
        ral.init_mst(np.tile([0, 0, 0, 1.0], (4, 1)), QQ, I, 1)
    assert e.value.code == capi.ERR_NOT_SPANNING
    with pytest.raises(capi.IrotavgError) as e:                 # assert(f>0) in the reference
        ral.init_mst(np.tile([0, 0, 0, 1.0], (4, 1)), QQ, I, 0)
    assert e.value.code == capi.ERR_BAD_ARG


def test_init_mst_order_dependence_matches_oracle():
    G = synth.make_graph(80, 500, 0.3, seed=9)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(G["I"]))
    I, QQ = G["I"][perm], G["QQ"][perm]
    flip = rng.random(len(I)) < 0.5                            # i > j edges take the backward branch
    I2 = np.where(flip[:, None], I[:, ::-1], I).astype(np.int32)
    QQ2 = np.where(flip[:, None], synth.qconj(QQ), QQ)
    Q = np.zeros((80, 4)); Q[:, 3] = 1
    Qa = Q.copy()
    ral.init_mst(Qa, QQ2, I2, 3)
    rc, Qb = O.init_mst(Q, QQ2, I2, 3)
    np.testing.assert_array_equal(Qa, Qb)


def test_make_A_matches_oracle_including_quirk():
    I = np.array([[0, 2], [3, 1], [2, 3], [0, 1], [4, 4], [4, 2]], dtype=np.int32)
    colptr, rowidx, vals = ral.make_A(5, 2, I)
    A = O.make_A(5, 2, I)
    np.testing.assert_array_equal(colptr, A.indptr)
    np.testing.assert_array_equal(rowidx, A.indices)
    np.testing.assert_array_equal(vals, A.data)
    with pytest.raises(capi.IrotavgError):                      # assert(n-f > 1)
        ral.make_A(3, 2, I[:1])


def test_compute_entry_points_fail_loudly_without_a_device():
    """No CPU fallback: without a HIP device graph creation must return IROTAVG_ERR_NO_DEVICE."""
    if capi.lib().irotavg_device_count() > 0:
        pytest.skip("a GPU is present")
    I = np.array([[0, 1]], dtype=np.int32)
    QQ = np.array([[0, 0, 0, 1.0]])
    with pytest.raises(capi.IrotavgError) as e:
        capi.Graph(I, QQ, 2, 1)
    assert e.value.code == capi.ERR_NO_DEVICE
    Q = np.array([[0, 0, 0, 1.0], [0, 0, 0, 1.0]])
    w = np.zeros(1)
    with pytest.raises(capi.IrotavgError) as e:
        ral.irls(QQ, I, None, ral.Geman_McClure, 0.1, Q, 1, 5, 1e-3, w)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.IrotavgError) as e:
        ral.quat_normalised(Q, 1)
    assert e.value.code == capi.ERR_NO_DEVICE


def test_bad_arguments():
    L = capi.lib()
    h = C.c_void_p()
    assert L.irotavg_graph_create(C.byref(h), 0, 2, 1, None, None, 0, None) == capi.ERR_BAD_ARG
    assert L.irotavg_graph_irls(None, 4, 0.1, 1, 1e-3, None, None, None) == capi.ERR_BAD_ARG
    # irotavg_window_solve: f and the edge endpoints index LDS unguarded inside the kernel, so they
    # are validated at the ABI (before any device is touched)
    I = np.array([[0, 1], [1, 2]], dtype=np.int32)
    QQ = np.tile([0, 0, 0, 1.0], (2, 1))
    Q = np.tile([0, 0, 0, 1.0], (3, 1))
    for bad_f in (-5, 3):
        with pytest.raises(capi.IrotavgError) as e:
            capi.window_solve(I, QQ, Q.copy(), bad_f)
        assert e.value.code == capi.ERR_BAD_ARG
    for bad in (-1, 3):
        Ib = I.copy(); Ib[1, 1] = bad
        with pytest.raises(capi.IrotavgError) as e:
            capi.window_solve(Ib, QQ, Q.copy(), 1)
        assert e.value.code == capi.ERR_BAD_ARG
    assert ral.parse_cost("geman-mcclure") == 4 and ral.parse_cost("L1.5") == 2
    with pytest.raises(ValueError):
        ral.parse_cost("nope")
