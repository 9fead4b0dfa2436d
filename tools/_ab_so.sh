#!/bin/bash
cp retto_amd/libretto_hip.so /tmp/new.so; cp retto_amd/libretto_hip_b0.so /tmp/old.so
for r in 1 2; do
  for v in new old; do
    cp /tmp/$v.so retto_amd/libretto_hip.so
    echo "== $v"
    python tools/layer_profile.py 32 3 2>&1 | grep -E "total profiled|gemm_pw/thin|gemm_ctc|gemm_neck|gemm_misc|gemm_cls"
  done
done
cp /tmp/new.so retto_amd/libretto_hip.so
