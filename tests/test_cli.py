"""The l1_irls-compatible driver (tools/l1_irls.cpp, ral/test.cpp's arguments and formats)."""
import os
import subprocess

import numpy as np
import pytest

from irotavg_amd import buildlib, capi, graphio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "ravg_input.txt")


def cli():
    if not os.path.exists(buildlib.CLI):
        buildlib.build()
    return buildlib.CLI


def test_usage_and_argument_errors():
    r = subprocess.run([cli()], capture_output=True, text=True)
    assert r.returncode == 255 and "Usage" in r.stderr       # exit(-1) in the reference
    r = subprocess.run([cli(), "/nonexistent/file"], capture_output=True, text=True)
    assert r.returncode == 255 and "Unable to open file" in r.stderr
    r = subprocess.run([cli(), FIX, "/tmp/_o.txt", "not-a-cost"], capture_output=True, text=True)
    assert r.returncode == 255 and "Unknown string" in r.stderr


def test_fails_loudly_without_a_device(tmp_path):
    if capi.lib().irotavg_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([cli(), FIX, str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert r.returncode == 255 and "no usable HIP device" in r.stderr
    assert not (tmp_path / "o.txt").exists()


@pytest.mark.gpu
def test_fixture_end_to_end_matches_golden(tmp_path):
    out = tmp_path / "l1_irls_out.txt"
    r = subprocess.run([cli(), FIX, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "L1-RA iterations = 1" in r.stdout and "IRLS  iterations = 2" in r.stdout
    Q, w = graphio.read_l1_irls_out(str(out), 1832)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "fixture_expected.npz"))
    assert len(w) == 3655
    assert synth.angular_distance(Q, gold["Q"]).max() < 1e-8
    np.testing.assert_allclose(w, gold["weights"], rtol=1e-8)


@pytest.mark.gpu
def test_all_arguments(tmp_path):
    out = tmp_path / "o.txt"
    r = subprocess.run([cli(), FIX, str(out), "L1", "3", "7", "2", "0.0005"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "cost: L1" in r.stdout and "sigma [deg]: 3" in r.stdout
    from oracle import oracle as O
    g = graphio.read_ravg_input(FIX)
    rc, Q0 = O.init_mst(g["Q"], g["QQ"], g["I"], 1)
    a = O.l1ra(g["QQ"], g["I"], Q0, 1, 2, 0.0005)
    b = O.irls(g["QQ"], g["I"], a["Q"], 1, 1, 3 * np.pi / 180, 7, 0.0005)
    Q, w = graphio.read_l1_irls_out(str(out), 1832)
    assert synth.angular_distance(Q, O.quat_normalised(b["Q"], 1)).max() < 1e-8
    np.testing.assert_allclose(w, b["weights"], rtol=1e-7)


@pytest.mark.gpu
def test_native_stream_driver_small():
    """tools/stream_bench.cpp (BASELINE config 5 driven by a native loop over the view-graph C ABI) at a small size:
    one JSON line, every window solved, the global re-solves on the loop closures ran, error at the noise level."""
    import json
    from irotavg_amd import buildlib
    buildlib.build()
    exe = os.path.join(os.path.dirname(buildlib.CLI), "stream_bench")
    r = subprocess.run([exe, "600", "900", "3", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["streamed_views"] == 900 and d["loop_closures"] == 3
    assert d["views_per_s"] > 1000
    assert d["mean_angular_error_rad"] < 0.03 and d["max_angular_error_rad"] < 0.1
    # five independent sessions in lock-step: their windows share one launch per step (irotavg_viewgraph_rot_avg_batch)
    r = subprocess.run([exe, "600", "900", "2", "1", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d5 = json.loads(r.stdout.strip().splitlines()[-1])
    assert d5["sessions"] == 5 and d5["loop_closures"] == 10
    assert d5["mean_angular_error_rad"] < 0.03 and d5["max_angular_error_rad"] < 0.1
