set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4base
timeout 300 python bench.py > gpurun_out/r4base/bench_final.json 2> gpurun_out/r4base/bench_final.err
timeout 1500 bash tools/profile_counters.sh gpurun_out/r4base/prof > gpurun_out/r4base/prof.log 2>&1
timeout 300 python bench.py --views 1000000 --edges 20000000 --steps 5 --warmup 1 --ramp 5 > gpurun_out/r4base/bench_1M20M.json 2> gpurun_out/r4base/bench_1M20M.err
timeout 200 python bench.py --views 10000 --edges 150000 > gpurun_out/r4base/bench_10k150k.json 2> gpurun_out/r4base/bench_10k.err
timeout 200 python bench.py --views 10000 --edges 150000 --p-loop 0.02 > gpurun_out/r4base/bench_10k150k_loop02.json 2>> gpurun_out/r4base/bench_10k.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4base/l1band -o l -- python $GRAFT_REPO_ROOT/tools/prof_case.py --what l1ra --reps 7 > $GRAFT_REPO_ROOT/gpurun_out/r4base/l1band.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 irotavg_amd/bin/stream_bench 50000 50000 10 0 1 > gpurun_out/r4base/stream_c4.json 2> gpurun_out/r4base/stream.err
timeout 200 python tools/time_global_resolve.py > gpurun_out/r4base/global_resolve.log 2>&1
ls -la gpurun_out/r4base
