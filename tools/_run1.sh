export SHAPES=1230432x240x240,307608x480x480,1230432x128x240
for nt in 0 1; do echo "RT_G32P_NT=$nt"; RT_G32P_NT=$nt python tools/bench_gemm.py 30 2>&1 | tail -3; done
