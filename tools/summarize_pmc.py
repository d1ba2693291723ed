"""Per-kernel summary of the rocprofv3 passes written by tools/profile_counters.sh:
average duration (kernel trace) and HBM traffic per launch from FETCH_SIZE / WRITE_SIZE.

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports one half of the bytes of wide coalesced
streaming reads, so the read side is doubled ("fetch_x2"); WRITE_SIZE is uncalibrated and taken
as is. traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch.
The summary is stamped with the workload string of the bench line the trace pass printed
(`_meta.workload`): bench.py only quotes `roofline.traffic` from a summary whose workload is the one it
is running (a stale number next to another size's algorithmic bytes is worse than null).
Usage: python tools/summarize_pmc.py gpurun_out/prof5 profiles/r02_pmc_summary.json
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("irh::", "")
    return n


def summarize(d):
    """d: directory with trace/t_kernel_stats.csv, fetch/f_counter_collection.csv, write/w_counter_collection.csv"""
    stats = {}
    for r in csv.DictReader(open(d + "/trace/t_kernel_stats.csv")):
        stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                       pct=float(r["Percentage"]))
    for tag, col in (("fetch", "f_counter_collection.csv"), ("write", "w_counter_collection.csv")):
        acc = defaultdict(list)
        for r in csv.DictReader(open("%s/%s/%s" % (d, tag, col))):
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if k in stats:
                # steady state: drop the smallest launches of a kernel (early-exit no-op launches)
                v = sorted(v)
                v = v[len(v) // 4:]
                stats[k][tag + "_kib"] = sum(v) / len(v)
    for k, s in stats.items():
        if "fetch_kib" in s and "write_kib" in s:
            s["traffic_bytes"] = 2 * s["fetch_kib"] * 1024 + s["write_kib"] * 1024
    return stats


def main(d, out):
    stats = summarize(d)
    meta = {}
    try:  # the bench line of the trace pass (tools/profile_counters.sh redirects stdout into trace.log)
        for line in open(d + "/trace.log"):
            if line.startswith("{") and '"config"' in line:
                b = json.loads(line)
                meta = dict(workload=b["config"]["workload"], pcg_rtol=b["config"].get("pcg_rtol"),
                            ms_per_step=b.get("ms_per_step"), value=b.get("value"))
    except Exception as e:
        meta = dict(error=str(e))
    stats["_meta"] = meta
    json.dump(stats, open(out, "w"), indent=1)
    stats.pop("_meta")
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["pct"])[:16]:
        print("%-34s calls %6d avg %9.1f us %5.1f%%  fetch %10.0f KiB write %10.0f KiB traffic %8.1f MB" % (
            k[:34], s["calls"], s["avg_us"], s["pct"], s.get("fetch_kib", -1), s.get("write_kib", -1),
            s.get("traffic_bytes", 0) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
