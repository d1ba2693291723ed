import os

import numpy as np
import pytest

from irotavg_amd import graphio, synth


def test_ravg_roundtrip_and_vertex_remap(tmp_path):
    G = synth.make_graph(30, 120, 0.2, seed=1)
    ids = np.arange(30) * 7 + 3                      # arbitrary ids are remapped to sorted rank
    p = tmp_path / "g.txt"
    graphio.write_ravg_input(str(p), ids[G["I"]], G["QQ"], G["Qgt"][:2], 30, 2)
    g = graphio.read_ravg_input(str(p))
    assert (g["m"], g["n"], g["f"], g["n_abs_read"]) == (120, 30, 2, 2)
    np.testing.assert_array_equal(g["I"], G["I"])
    np.testing.assert_allclose(g["QQ"], G["QQ"], rtol=0, atol=0)
    np.testing.assert_array_equal(g["Q"][:2], G["Qgt"][:2])
    assert not g["Q"][2:].any()


def test_ravg_errors(tmp_path):
    p = tmp_path / "bad.txt"
    p.write_text("2 3 1\n0 1 1 0 0 0\n")
    with pytest.raises(graphio.GraphFileError):
        graphio.read_ravg_input(str(p))                      # inconsistent number of connections
    p.write_text("1 2 1\n0 1 1 0 0 0\n")
    with pytest.raises(graphio.GraphFileError):
        graphio.read_ravg_input(str(p))                      # fewer than f absolute rotations
    p.write_text("1 3 1\n0 1 1 0 0 0\n1 0 0 0\n")
    with pytest.raises(graphio.GraphFileError):
        graphio.read_ravg_input(str(p))                      # n != max(j)+1


def test_output_format_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    Q = rng.normal(size=(5, 4)); w = rng.random(7)
    p = tmp_path / "o.txt"
    graphio.write_l1_irls_out(str(p), Q, w)
    Q2, w2 = graphio.read_l1_irls_out(str(p), 5)
    np.testing.assert_array_equal(Q, Q2)
    np.testing.assert_array_equal(w, w2)
    first = open(p).readline().split()
    assert float(first[0]) == Q[0, 3]                        # file order is w x y z


@pytest.mark.parametrize("n,m,p", [(1000, 15000, 0.0), (1000, 15000, 0.02), (500, 2000, 0.3)])
def test_generator_topology(n, m, p):
    G = synth.make_graph(n, m, p, seed=0)
    I = G["I"]
    assert len(I) == m and (I[:, 0] < I[:, 1]).all()
    assert (np.diff(I[:, 1]) >= 0).all()                     # grouped by the newer view
    assert G["is_loop"].sum() == round(p * m)
    band = I[~G["is_loop"]]
    assert (band[:, 1] - band[:, 0]).max() <= G["w"] + 1
    loops = I[G["is_loop"]]
    if len(loops):
        assert (loops[:, 1] - loops[:, 0]).min() > G["w"] + 1
        assert len(np.unique(loops[:, 0].astype(np.int64) * n + loops[:, 1])) == len(loops)
    assert G["is_outlier"].sum() == round(0.05 * G["is_loop"].sum())
    np.testing.assert_allclose(np.linalg.norm(G["QQ"], axis=1), 1, atol=1e-12)
    G2 = synth.make_graph(n, m, p, seed=0)
    np.testing.assert_array_equal(G["QQ"], G2["QQ"])         # seeded


def test_generator_noise_model():
    G = synth.make_graph(300, 3000, 0.1, sigma_n=0.01, p_out=0.0, seed=3)
    d = synth.qmul(G["QQ"], synth.qmul(G["Qgt"][G["I"][:, 0]], synth.qconj(G["Qgt"][G["I"][:, 1]])))
    ang = synth.angular_distance(d, np.tile([0, 0, 0, 1.0], (len(d), 1)))
    assert 0.012 < ang.mean() < 0.020                        # |N(0, 0.01^2 I3)| has mean ~0.016
