"""Side-by-side of tools/layer_profile.py outputs: python tools/cmp_layers.py prefix file_A file_B ... (rows whose family starts with prefix)"""
import re, sys, collections
pre, files = sys.argv[1], sys.argv[2:]
rows = collections.OrderedDict()
for i, f in enumerate(files):
    for l in open(f):
        m = re.match(r"\s*([\d.]+) (\S+)\s+(\S*)\s+(\d+)\s+([\d.]+)", l)
        if not m or not m.group(2).startswith(pre):
            continue
        rows.setdefault((m.group(2), m.group(3), int(m.group(4))), {})[i] = float(m.group(1))
tot = [0.0] * len(files); best = 0.0
for k, v in rows.items():
    print("%-22s %-30s %3d " % k + " ".join("%7.3f" % v.get(i, 0) for i in range(len(files))))
    for i in range(len(files)):
        tot[i] += v.get(i, 0)
    best += min(v.values())
print("total", [round(t, 2) for t in tot], "best-of", round(best, 2))
