set -u
# The round's judged numbers in one go (GPU box, from the repo root; `gpurun -- bash tools/collect_profiles.sh`): the default
# bench line with its own rocprofv3 passes (kernel trace + FETCH_SIZE + WRITE_SIZE, gpurun_out/bench_profile/), the other
# sizes, l1ra under the kernel trace, config 5, the GPU test suite. Everything lands in gpurun_out/final;
# tools/publish_profiles.sh NN copies the summaries to profiles/rNN_*.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
cp -r gpurun_out/bench_profile $OUT/bench_profile 2>/dev/null
timeout 400 python bench.py --views 1000000 --edges 20000000 --steps 20 --warmup 2 --ramp 5 --no-pmc > $OUT/bench_1M20M.json 2> $OUT/bench_1M20M.err
timeout 300 python bench.py --views 10000 --edges 150000 --no-pmc > $OUT/bench_10k150k.json 2> $OUT/bench_10k.err
timeout 300 python bench.py --views 10000 --edges 150000 --p-loop 0.02 --steps 50 --no-pmc > $OUT/bench_10k150k_loop02.json 2>> $OUT/bench_10k.err
timeout 600 python bench.py --p-loop 0.02 --steps 20 > $OUT/bench_loop02.json 2> $OUT/bench_loop02.err
cp -r gpurun_out/bench_profile $OUT/bench_profile_loop02 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/l1band -o l -- python $GRAFT_REPO_ROOT/tools/prof_case.py --what l1ra --reps 7 > $GRAFT_REPO_ROOT/$OUT/l1band.log 2>&1
for nc in 30 100; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/cl$nc -o c -- python $GRAFT_REPO_ROOT/tools/dev/prof_closures.py 100000 2000000 $nc > $GRAFT_REPO_ROOT/$OUT/cl$nc.log 2>&1; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/k1cold -o k -- python $GRAFT_REPO_ROOT/tools/dev/k1_cold_cache.py > $GRAFT_REPO_ROOT/$OUT/k1cold.log 2>&1
cd $GRAFT_REPO_ROOT
for c in neartree_seed603_case163 neartree_seed501_case196; do timeout 300 python tools/referee.py --npz tests/golden/$c.npz --gpu --json $OUT/referee_$c.json > $OUT/referee_$c.txt 2>&1; done
timeout 600 python tools/dev/inexact_vs_oracle.py > $OUT/inexact_vs_oracle.jsonl 2>&1
for rep in 1 2 3; do timeout 300 irotavg_amd/bin/stream_bench 50000 50000 10 0 1 1; done > $OUT/stream_c4.json 2> $OUT/stream.err
timeout 300 irotavg_amd/bin/stream_bench 50000 50000 10 0 1 0 > $OUT/stream_c4_no_prepare.json 2>> $OUT/stream.err
for s in 1 8 64; do timeout 300 irotavg_amd/bin/stream_bench 5000 5000 0 0 $s 0 | cut -c1-700; done > $OUT/stream_sessions.jsonl 2>> $OUT/stream.err
timeout 200 python tools/time_global_resolve.py > $OUT/global_resolve.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
ls -la $OUT
