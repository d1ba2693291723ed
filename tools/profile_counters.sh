#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace + stats, then the HBM counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, PMC slots).
# Every pass runs under `timeout`: rocprofv3 has been seen to sit for minutes after "tool finalization".
# Usage (on the GPU box, from the repo root): bash tools/profile_counters.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/prof}; shift || true
R=$PWD
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extra "$@" > "$R/$OUT/trace.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/fetch" -o f -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-extra "$@" > "$R/$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/write" -o w -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-extra "$@" > "$R/$OUT/write.log" 2>&1
cd "$R"
ls -R "$OUT" | head -30
