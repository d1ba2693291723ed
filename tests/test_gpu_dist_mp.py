"""The sharded path with more than one PROCESS (dist.hip had only ever run in one):
torch.distributed.run starts 2 ranks that share the one GPU of a test box, each holding one shard.

* host-staged wire (irotavg_transport over gloo): must pass -- this is the real multi-process control
  flow of dist.hip (per-rank plans, halo exchanges, reductions that steer every rank identically);
* RCCL wire: ncclCommInitRank with two ranks on ONE device. RCCL is entitled to refuse that
  ("Duplicate GPU detected"); then the test is skipped with that reason. On a multi-GPU node the same
  worker runs one rank per GPU (tools/dist_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(wire, nproc=2, port=29611, timeout=420, extra=()):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_worker.py"),
           "--wire", wire] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_two_processes_host_staged_wire():
    r = run_worker("hosted")
    assert r.returncode == 0 and "DIST_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("nproc", [2, 3])
def test_processes_host_staged_wire_sharded_direct_solver(nproc):
    """a view sequence without loop closures: every rank reduces its range, ONE gather per linear solve, the top
    levels solved on every rank (bcr_dist); l1ra then irls against the single-GPU handle"""
    r = run_worker("hosted", nproc=nproc, port=29619 + nproc, extra=["--p-loop", "0", "--views", "9000", "--edges", "90000",
                                                                     "--expect-direct"])
    assert r.returncode == 0 and "DIST_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("nproc,closures", [(2, 12), (3, 150)])
def test_processes_host_staged_wire_sharded_direct_solver_with_closures(nproc, closures):
    """a view sequence WITH loop closures between processes (round 5): besides the separators' gather the ranks sum one
    buffer (the closures' columns on the separators, every rank's share of the Woodbury system) through the hosted
    wire's all-reduce; l1ra then irls against the single-GPU handle"""
    r = run_worker("hosted", nproc=nproc, port=29629 + nproc, extra=["--p-loop", "0", "--views", "9000", "--edges", "90000",
                                                                     "--closures", str(closures), "--expect-direct"])
    assert r.returncode == 0 and "DIST_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_two_processes_host_staged_wire_guarded_closure_solve():
    """a dead pivot of the band part (Talwar cuts a view off all its band neighbours, one closure holds it) between two
    processes: every rank reads the summed count, the solve is repeated by conjugate gradients whose operator is the
    sharded SpMV (halo exchange of p through the wire), whose preconditioner is the regularised sharded direct solve and
    whose three sums per iteration are combined over the processes (bcr_dist_checked)"""
    r = run_worker("hosted", nproc=2, port=29641, extra=["--p-loop", "0", "--views", "9000", "--edges", "90000", "--closures", "6",
                                                         "--cut-stretch", "--expect-direct"])
    assert r.returncode == 0 and "DIST_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_two_processes_rccl_on_one_device():
    try:
        r = run_worker("rccl", port=29613, timeout=300)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL with two ranks on one device did not come up within the time limit")
    if r.returncode != 0:
        tail = (r.stdout + r.stderr)[-1500:]
        pytest.skip("RCCL refused two ranks on one device (expected on a 1-GPU box): ..." + tail[-400:])
    assert "DIST_WORKER_OK" in r.stdout


def test_bench_runs_with_two_ranks():
    """bench.py's own N > 1 branch (sharding, timing protocol, max over ranks, the JSON line), two ranks
    sharing the box's one GPU (IROTAVG_BENCH_SHARE_GPU=1). RCCL refuses two ranks on one device, so this
    exercises both answers to "the library's communicator cannot be formed": the refusal (default) and, with
    --allow-hosted, the fall-back every rank takes together -- the same sharded solver over the host-staged
    transport -- with the `dist` record a first multi-GPU run diagnoses itself by."""
    import json
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["IROTAVG_BENCH_SHARE_GPU"] = "1"
    common = ["--views", "20000", "--edges", "300000", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extra", "--no-pmc"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "2"] + common
    # (1) without --allow-hosted a run whose RCCL communicator cannot be formed REFUSES: non-zero exit, one JSON line
    # that says why and carries no value -- a first hardware run can never silently measure gloo
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    refused = json.loads(lines[0])
    assert refused["value"] is None and "refusing" in refused["error"] and refused["dist"]["wire"] == "none"
    # (2) with it: the same sharded solver over the host-staged wire, and the line says so in machine-readable form
    cmd[cmd.index("29617")] = "29618"
    r = subprocess.run(cmd + ["--allow-hosted"], capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    two = json.loads(lines[0])
    d = two["dist"]
    assert d["wire"] == "host-staged" and d["ncclCommCount"] == 0 and d["world"] == 2 and d["hosted_allowed"] is True
    assert d["valid_scaling_measurement"] is False           # a hosted or GPU-sharing run is never a scaling figure
    assert d["sharded_solver"] == "direct" and d["phases_iterations"] > 0
    assert d["matches_single_gpu"] is True and d["single_gpu_iters"] == two["iters_to_converge"] and d["max_rel_score_diff"] < 1e-6
    ph = d["phases_us_per_iteration"]
    assert set(ph) == {"local_edge_kernels_and_assembly", "local_reductions", "gather_of_separators", "closure_sum",
                       "separator_system_and_ways_back", "halo_of_the_step", "weights_and_rotation_update",
                       "score_allreduce"}
    assert ph["local_reductions"] > 0 and ph["gather_of_separators"] > 0 and ph["halo_of_the_step"] > 0
    assert ph["closure_sum"] == 0.0                           # no closures in this graph
    assert "ms_per_step_by_gpus" in d["expected"] or d["expected"]["workload"] is None
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=420, cwd=ROOT)
    one = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["value"] > 0
    assert two["iters_to_converge"] == one["iters_to_converge"]
    assert "2 contiguous ranges" in two["config"]["parallelism"]
    np_ = __import__("numpy")
    np_.testing.assert_allclose(two["final_scores"], one["final_scores"], rtol=1e-6)
