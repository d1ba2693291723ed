"""Builds libirotavg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
CLI = os.path.join(HERE, "bin", "l1_irls")
LIB = os.path.join(HERE, "libirotavg_hip.so")
SOURCES = ["build.cpp", "gbuild.hip", "solver.hip", "cgcg.hip", "dense.hip", "bcr.hip", "l1pd.hip", "capi.cpp", "viewgraph.cpp", "dist.hip", "window.hip", "resident.hip"]
HEADERS = ["common.hpp", "graph.hpp", "kernels.hpp", "../../include/irotavg_hip.h"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    extra = [os.path.join(os.path.dirname(HERE), "tools", "l1_irls.cpp"),
             os.path.join(os.path.dirname(HERE), "tools", "stream_bench.cpp"),
             os.path.join(os.path.dirname(HERE), "include", "irotavg", "l1_irls.hpp")]
    if not os.path.exists(CLI) or not os.path.exists(os.path.join(os.path.dirname(CLI), "stream_bench")) or any(os.path.getmtime(e) > t for e in extra if os.path.exists(e)):
        return True
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS
               if os.path.exists(os.path.join(CSRC, f)))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    for src in SOURCES:
        obj = os.path.join(CSRC, src + ".o")
        objs.append(obj)
        # an object that is newer than its source and every header is kept (force=True recompiles everything)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-pthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(verbose)
    return LIB


CLI = os.path.join(HERE, "bin", "l1_irls")


def build_cli(verbose=False):
    """The `l1_irls`-compatible driver (tools/l1_irls.cpp over include/irotavg/l1_irls.hpp)."""
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    src = os.path.join(os.path.dirname(HERE), "tools", "l1_irls.cpp")
    cmd = ["g++", "-O2", "-std=c++11", "-DIROTAVG_SHIM_NO_EIGEN", src, "-o", CLI, "-L" + HERE,
           "-lirotavg_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the native driver of BASELINE.json config 5 (tools/stream_bench.cpp over the C ABI)
    src = os.path.join(os.path.dirname(HERE), "tools", "stream_bench.cpp")
    cmd = ["g++", "-O2", "-std=c++11", src, "-o", os.path.join(os.path.dirname(CLI), "stream_bench"), "-L" + HERE,
           "-lirotavg_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
