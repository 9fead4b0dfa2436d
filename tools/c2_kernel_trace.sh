#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a small-batch workload (serial calls) and the per-kernel table -- launches per call,
# average duration, share of the call's kernel time.  This is the view that showed the batch-independent latency chains of round 5
# ("Small-batch latency").
#   gpurun -- 'bash tools/c2_kernel_trace.sh [rows] [bench args]'      default: 30 rows, "--workload c2" (one 960 x 960 page per call);
#   e.g. 'bash tools/c2_kernel_trace.sh 40 --pages 1' = one page with its 32 text lines through det + cls + rec
rows=${1:-30}; shift
args=${*:---workload c2}
steps=200; calls=$((steps + 20 + 2 * steps + 3))
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c2_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2_trace -o t -- python3 bench.py $args --steps $steps --warmup 20 --no-cpu-baseline --no-c5 --lanes 1 --inflight 1 --repeat 0 > gpurun_out/c2_trace.log 2>&1
grep '^{' gpurun_out/c2_trace.log | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('bench:', j['value'], j['unit'], j['ms_per_step'], 'ms per call; networks', {k:v for k,v in j['networks'].items() if k.endswith('_ms')})"
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/c2_trace/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
# calls of the whole process: warmup + timed + the serial profile passes; normalise by the most frequent once-per-call kernel
import collections
per=collections.Counter(int(r["Calls"]) for r in rows).most_common(1)[0][0]
print("kernel ms per call %.3f, launches per call %.1f (a once-per-call kernel ran %d times)" % (tot/per/1e6, sum(int(r["Calls"]) for r in rows)/per, per))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:$rows]:
    print("%6.2f%% per call %5.1f avg %8.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"])/per, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
