"""GPU-busy fraction of a production run from a rocprofv3 kernel trace: the union of all kernel intervals over the span of
the densest part of the trace (the timed steps).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/busy -o t -- python3 bench.py --no-cpu-baseline --no-c5 --repeat 0
    python tools/gpu_busy.py gpurun_out/busy/.../t_kernel_trace.csv [window_seconds]
"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
t_end = max(e for _s, e, _n in rows)
# the window: the `win` seconds that hold the most kernel launches (the timed region is the densest stretch of the run)
starts = [s for s, _e, _n in rows]
best, j = (0, 0), 0
W = int(win * 1e9)
for i, s in enumerate(starts):
    while starts[j] < s - W:
        j += 1
    if i - j > best[0]:
        best = (i - j, s)
w1 = best[1]; w0 = w1 - W
busy, cur_s, cur_e = 0, None, None
conc = 0
for s, e, _n in rows:
    if e <= w0 or s >= w1:
        continue
    s, e = max(s, w0), min(e, w1)
    conc += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
print("window %.3f s, %d kernels: GPU busy (some kernel running) %.1f %%, mean kernels in flight %.2f" % (
    W / 1e9, best[0], 100.0 * busy / W, conc / W))
