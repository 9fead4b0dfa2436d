#!/bin/bash
# A/B of two builds on one box: retto_amd/libretto_hip.so (new) vs retto_amd/libretto_hip_b0.so (old form), alternating
cp retto_amd/libretto_hip.so /tmp/new.so; cp retto_amd/libretto_hip_b0.so /tmp/old.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
for r in 1 2; do
  for v in new old; do
    cp /tmp/$v.so retto_amd/libretto_hip.so
    echo "== $v"
    python tools/layer_profile.py 32 3 2>&1 | grep -E "total profiled|se_pool_fc|global_mean"
    python tools/layer_profile.py 1 20 0 2>&1 | grep -E "total profiled|se_pool_fc"
    bash tools/_ab.sh ":--steps 8 --warmup 3" ":--workload c2"
  done
done
cp /tmp/new.so retto_amd/libretto_hip.so
