#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh <tag>'): rocprofv3 kernel stats of the default bench.py (C3) and
# of --workload c5, the three PMC passes of each (kept apart: they do not fit one pass on gfx950), tools/pmc_summary.py, and the
# per-layer tables.  Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/.
set -u
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C3="--steps 4 --warmup 3 --no-cpu-baseline --no-c5 --lanes 1 --inflight 1"
C5="--workload c5 --pages 32 --steps 2 --warmup 3 --no-cpu-baseline --lanes 1 --inflight 1"   # (32 pages: the leg the default bench.py run appends as `c5`)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_trace -o t -- python3 bench.py $C3 > $out/c3_bench.log 2> $out/c3_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5_trace -o t -- python3 bench.py $C5 > $out/c5_bench.log 2> $out/c5_bench.err
for w in c3 c5; do
  if [ $w = c3 ]; then F="--steps 2 --warmup 2 --no-cpu-baseline --no-c5 --lanes 1 --inflight 1"; else F="--workload c5 --pages 32 --steps 1 --warmup 3 --no-cpu-baseline --lanes 1 --inflight 1"; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${w}_fetch -o f -- python3 bench.py $F > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${w}_write -o w -- python3 bench.py $F > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/${w}_sq -o s -- python3 bench.py $F > /dev/null 2>&1
done
python3 tools/pmc_summary.py $out/c3_fetch/f_counter_collection.csv $out/c3_write/w_counter_collection.csv $out/pmc_traffic_c3.json $out/c3_sq/s_counter_collection.csv '{"workload": "c3", "pages": 32, "size": 960, "lines": 32, "dtype": "f32", "models": "mobile"}' > $out/pmc_c3.txt
python3 tools/pmc_summary.py $out/c5_fetch/f_counter_collection.csv $out/c5_write/w_counter_collection.csv $out/pmc_traffic_c5.json $out/c5_sq/s_counter_collection.csv '{"workload": "c5", "pages": 32, "size": 960, "lines": 32, "dtype": "f16", "models": "server"}' > $out/pmc_c5.txt
# det network alone: HBM bytes per pass from the counters (north_star: rocprof-reported HBM GB/s of the DBNet backbone)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/det_fetch -o f -- python3 tools/layer_profile.py 32 1 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/det_write -o w -- python3 tools/layer_profile.py 32 1 0 > /dev/null 2>&1
python3 tools/pmc_det_bytes.py $out/det_fetch/f_counter_collection.csv $out/det_write/w_counter_collection.csv $out/pmc_det_c3.json > $out/pmc_det.txt 2>&1
RT_PROFILE_DETAIL=1 python3 tools/layer_profile.py 32 3 > $out/c3_layers.txt 2>&1
RT_PROFILE_DETAIL=1 python3 tools/layer_profile.py 32 3 0 > $out/c3_det_layers.txt 2>&1
RT_GEMM_SPLIT=1 RT_PROFILE_DETAIL=1 python3 tools/layer_profile.py 32 3 > $out/c3_layers_split.txt 2>&1   # (round 6: the opt-in split-bf16 kernels)
SHAPES=1230432x240x240,307608x480x480,1230432x128x240 ITERS=10 python3 tools/bench_gemm_split.py 30 40 > $out/gemm_split.txt 2>&1
WORKLOAD=c5 RT_PROFILE_DETAIL=1 python3 tools/layer_profile.py 32 3 > $out/c5_layers.txt 2>&1
cp $out/c3_trace/t_kernel_stats.csv $out/c3_kernel_stats.csv 2>/dev/null
cp $out/c5_trace/t_kernel_stats.csv $out/c5_kernel_stats.csv 2>/dev/null
ls $out
# one bench line per other workload of the final build (C2: det only, one page per call; C4: mixed page sizes) and the default run
python3 bench.py --workload c2 --no-cpu-baseline --no-c5 > $out/bench_c2.json 2> $out/bench_c2.err
python3 bench.py --workload c4 --no-cpu-baseline --no-c5 > $out/bench_c4.json 2> $out/bench_c4.err
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
ls $out
