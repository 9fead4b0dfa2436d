#!/bin/bash
for o in 1 2 4 1 2 4; do
  echo "== RT_GEMM_OCC=$o"
  RT_GEMM_OCC=$o python tools/layer_profile.py 32 3 2>&1 | grep -E "total profiled|gemm_neck|gemm_pw/thin|gemm_misc|gemm_cls|gemm_ctc" 
  RT_GEMM_OCC=$o python tools/layer_profile.py 1 20 0 2>&1 | grep -E "total profiled"
done
