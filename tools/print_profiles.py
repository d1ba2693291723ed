"""Prints the figures of a tools/collect_profiles.sh run (gpurun_out/final or a directory given) that the docs quote."""
import json, sys, glob
D = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
def load(f):
    for line in open(f):
        line = line.strip()
        if line.startswith('{'):
            return json.loads(line)
d = load(D + '/bench_final.json')
print('headline', d['value'], d['ms_per_step'], 'throttled', d.get('host_throttled_usec'), d.get('host_throttled_usec_timed_region'))
r = d['roofline']
print('roofline algo', r['algorithmic_bytes'], 'ms', r['ms_per_launch'], 'frac', r['frac'], 'insitu', r['ms_per_launch_in_situ'], r.get('frac_in_situ'), 'traffic', r['traffic'], 'mfma', r['mfma']['achieved'], r['mfma']['frac'])
for k, v in r['by_launch'].items():
    print('  ', k, {kk: v[kk] for kk in v if kk in ('ms', 'ms_in_situ', 'frac', 'traffic', 'algorithmic_bytes')})
for k in d:
    if k.startswith('roofline_') and isinstance(d[k], dict):
        v = d[k]; print(k, {kk: v[kk] for kk in v if not isinstance(v[kk], (str, dict, list))})
for k in d:
    if k.startswith('also') and isinstance(d[k], dict):
        v = d[k]; print(k, json.dumps({kk: v[kk] for kk in v if not isinstance(v[kk], (str, dict, list))})[:420])
print('cpu', d['cpu_baseline']['value'])
for f in ['bench_1M20M', 'bench_10k150k', 'bench_10k150k_loop02', 'bench_loop02']:
    e = load(D + '/%s.json' % f)
    a = e.get('also_l1ra_then_irls', {})
    print(f, e['value'], e['ms_per_step'], a.get('l1ra_ms_per_outer_iteration'), a.get('l1ra_ms_min_max'), a.get('edge_updates_per_s_whole_pipeline'), (e.get('cpu_baseline') or {}).get('value'))
for f in ['stream_c4.json', 'stream_c4_no_prepare.json']:
    for l in open(D + '/' + f):
        if l.startswith('{'):
            x = json.loads(l); print(f, x['views_per_s'], x['local_rotavg_ms_mean'], x['local_rotavg_ms_p99'], x['global_rotavg_ms_mean'], x.get('global_rotavg_ms'))
for l in open(D + '/stream_sessions.jsonl'):
    if l.startswith('{'):
        x = json.loads(l); print('sessions', x['sessions'], x['views_per_s'])
for l in open(D + '/global_resolve.log'):
    if 'ms per call' in l: print(l.strip()[:160])
for l in open(D + '/pytest_gpu.log', errors='replace'):
    if 'passed' in l or 'failed' in l: print(l.strip())
