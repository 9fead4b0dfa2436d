# A/B: recognition launch-group size (input pixels per group) on the default bench workload
for px in 8000000 16000000 24000000 48000000 400000000; do
  echo -n "RT_REC_GROUP_PX=$px: "
  RT_REC_GROUP_PX=$px timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
done
