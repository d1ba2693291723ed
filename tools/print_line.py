import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"])
