#!/bin/bash
# FETCH_SIZE calibration for the slice-wise reads of the thin LCNetV3 kernels (tools/scratch/fetch_calib.hip); runs on the GPU box
out=gpurun_out/${1:-fetch_calib}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- ./tools/scratch/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $out/r -o r -- ./tools/scratch/fetch_calib > /dev/null 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel".ljust(40), "FETCH_SIZE [KiB] / 2^20 (1.0 = every byte counted)   RDREQ x 64 B / 2^30   32B share   BUBBLE x 128 B / 2^30")
for k in sorted(d):
    c = d[k]
    avg = lambda n: sum(c[n]) / max(1, len(c[n])) if n in c else float('nan')
    print(k.ljust(40), "%.3f" % (avg("FETCH_SIZE") / 2**20), " " * 30, "%.3f" % (avg("TCC_EA0_RDREQ_sum") * 64 / 2**30), "   %.3f" % (avg("TCC_EA0_RDREQ_32B_sum") / max(1.0, avg("TCC_EA0_RDREQ_sum"))), "   %.3f" % (avg("TCC_BUBBLE_sum") * 128 / 2**30))
PY
