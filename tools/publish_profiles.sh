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
for nc in 30 100; do [ -f $F/cl$nc/c_kernel_stats.csv ] && cp $F/cl$nc/c_kernel_stats.csv ${P}_closures_kernel_stats_$nc.csv; done
for c in neartree_seed603_case163 neartree_seed501_case196; do [ -f $F/referee_$c.json ] && cp $F/referee_$c.json ${P}_referee_$c.json; done
[ -f $F/inexact_vs_oracle.jsonl ] && grep -a '^{' $F/inexact_vs_oracle.jsonl > ${P}_inexact_vs_oracle.jsonl
[ -f $F/k1cold/k_kernel_trace.csv ] && python - $F/k1cold/k_kernel_trace.csv ${P}_k1_cold_cache.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_edge_residual' in r.get('Kernel_Name', '')]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
cold, warm = d[:30], d[30:60]
open(sys.argv[2], 'w').write(
    "k_edge_residual at 100k / 2M (131.2 MB algorithmic), tools/dev/k1_cold_cache.py under rocprofv3 --kernel-trace\n"
    "cold (each launch behind a 1 GiB fill: nothing of the working set in the 256 MiB Infinity Cache): mean %.2f us, median %.2f us -> %.3f of 8 TB/s\n"
    "warm (back to back: the 131 MB working set stays in the cache):                                  mean %.2f us, median %.2f us -> %.3f of 8 TB/s\n"
    % (sum(cold) / len(cold), sorted(cold)[len(cold) // 2], 131.2e6 / (sorted(cold)[len(cold) // 2] * 1e-6) / 8e12,
       sum(warm) / len(warm), sorted(warm)[len(warm) // 2], 131.2e6 / (sorted(warm)[len(warm) // 2] * 1e-6) / 8e12))
PY
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
