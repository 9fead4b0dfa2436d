"""One synchronous C2 call (det only, one 960 x 960 page) as a kernel table: name, start offset, duration, gap to the previous kernel.
Input: the kernel trace of `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --workload c2 --lanes 1 --inflight 1
--steps 50 --warmup 20 --no-cpu-baseline`; picks a call in the middle of the timed region (calls are delimited by the det stem)."""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return re.sub(r"rt::(nn|pp|nh)::", "", n)[:64]
names = [short(r["Kernel_Name"]) for r in rows]
stems = [i for i, n in enumerate(names) if n.startswith("k_stem_mfma<1>")]
a, b = stems[len(stems) // 2], stems[len(stems) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy = 0.0
prev = None
print("%3s %-64s %9s %8s %7s %6s" % ("#", "kernel", "start us", "dur us", "gap us", "WGs"))
for k, i in enumerate(range(a, b)):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    wg = 1
    for d in "XYZ":
        wg *= max(1, int(rows[i]["Grid_Size_" + d]) // max(1, int(rows[i]["Workgroup_Size_" + d])))
    print("%3d %-64s %9.1f %8.1f %7.1f %6d" % (k, names[i], (s - t0) / 1e3, (e - s) / 1e3, gap, wg))
    busy += (e - s) / 1e3
    prev = e
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print("call to call %.1f us, %d launches, kernels busy %.1f us (%.0f %%)" % (span, b - a, busy, 100 * busy / span))
# all calls: distribution of the call-to-call time and of the busy share
spans = []
for x, y in zip(stems[:-1], stems[1:]):
    spans.append((int(rows[y]["Start_Timestamp"]) - int(rows[x]["Start_Timestamp"])) / 1e3)
spans.sort()
print("call-to-call over %d calls: min %.1f median %.1f p90 %.1f us" % (len(spans), spans[0], spans[len(spans) // 2], spans[int(len(spans) * 0.9)]))
