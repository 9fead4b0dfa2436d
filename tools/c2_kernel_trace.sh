#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the C2 workload (one 960 x 960 page per call, serial) and the per-kernel table --
# launches per call, average duration, share of the call's kernel time.  This is the view that showed the batch-independent
# latency chains of round 5 ("Small-batch latency").  Usage: gpurun -- 'bash tools/c2_kernel_trace.sh [rows]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c2_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2_trace -o t -- python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-c5 --lanes 1 --inflight 1 > gpurun_out/c2_trace.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/c2_trace/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per call", tot/715/1e6, "launches per call", sum(int(r["Calls"]) for r in rows)/715)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:${1:-30}]:
    print("%6.2f%% per call %5.1f avg %8.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"])/715, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
