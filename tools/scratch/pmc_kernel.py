"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python pmc_kernel.py <substr> file.csv ..."""
import csv, sys, collections
sub = sys.argv[1]
for fn in sys.argv[2:]:
    agg = collections.defaultdict(lambda: [0.0, 0])
    try:
        rows = csv.DictReader(open(fn))
    except OSError as e:
        print(fn, e); continue
    for r in rows:
        k = r["Kernel_Name"]
        if sub not in k: continue
        k = k.split("(")[0][-60:]
        a = agg[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print("%-62s %-34s %14.0f  (n=%d)" % (k, c, v / n, n))
