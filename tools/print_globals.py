import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['views_per_s'], d['global_rotavg_ms_mean'], d.get('global_rotavg_ms'))
