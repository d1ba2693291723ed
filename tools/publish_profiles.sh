# copies the files of a tools/collect_profiles.sh run (gpurun_out/final) to profiles/rNN_* (NN = $1, default 06)
R=${1:-06}; F=gpurun_out/final; P=profiles/r${R}
if [ ! -f $F/bench_final.json ]; then echo "no run in $F"; exit 1; fi
cp $F/bench_final.json ${P}_bench_final.json; cp $F/bench_1M20M.json ${P}_bench_1M20M.json
cp $F/bench_10k150k.json ${P}_bench_10k150k.json; cp $F/bench_10k150k_loop02.json ${P}_bench_10k150k_loop02.json
cp $F/bench_loop02.json ${P}_bench_100k2M_loop02.json
cp $F/bench_profile/trace/t_kernel_stats.csv ${P}_bench_kernel_stats.csv; cp $F/bench_profile/pmc_summary.json ${P}_pmc_summary.json
cp $F/bench_profile_loop02/trace/t_kernel_stats.csv ${P}_pcg_kernel_stats_loop02.csv; cp $F/bench_profile_loop02/pmc_summary.json ${P}_pcg_pmc_summary_loop02.json
cp $F/l1band/l_kernel_stats.csv ${P}_kernel_stats_l1band.csv
cp $F/stream_c4.json ${P}_incremental_c4.json; cp $F/stream_c4_no_prepare.json ${P}_incremental_c4_no_prepare.json
cp $F/stream_sessions.jsonl ${P}_stream_sessions.jsonl; cp $F/global_resolve.log ${P}_global_resolve_75k.txt
grep -a "passed" $F/pytest_gpu.log > ${P}_pytest_gpu.txt
python - ${P} <<'PY'
import json, sys
P = sys.argv[1]
s = json.load(open(P + '_pmc_summary.json'))
meta = s.pop('_meta', {})
lines = ["workload: %s" % meta.get('workload', '(see the bench line)'),
         "rocprofv3 passes launched by bench.py itself (kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE); traffic = 2*FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section)", ""]
for k, v in sorted(s.items(), key=lambda kv: -kv[1].get('pct', 0))[:16]:
    lines.append("%-46s calls %5d avg %8.2f us %5.1f%%  fetch %9.0f KiB write %9.0f KiB traffic %8.2f MB" % (
        k[:46], v['calls'], v['avg_us'], v['pct'], v.get('fetch_kib', 0), v.get('write_kib', 0), v.get('traffic_bytes', 0) / 1e6))
open(P + '_pmc_table.txt', 'w').write("\n".join(lines) + "\n")
PY
