#!/bin/bash
# Compiles the kernel files that hold inline-asm vector-memory instructions to assembly (device only, gfx950) and scans them with
# tools/asm_hazard_scan.py for a VALU-written SGPR (v_readlane: the restore of a spilled scalar) read by an asm VMEM instruction
# fewer than 5 wait states later -- the hazard behind round 6's memory faults.  ~2.5 minutes on 8 cores.
set -u
cd "$(dirname "$0")/../retto_amd/csrc"
out=${TMPDIR:-/tmp}/rt_asm_scan; mkdir -p $out
for f in nn_gemm_dma nn_gemm_split nn_f16_dma nn_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -S $f.hip -o $out/$f.s 2>/dev/null &
done
wait
python3 ../../tools/asm_hazard_scan.py $out/nn_gemm_dma.s $out/nn_gemm_split.s $out/nn_f16_dma.s $out/nn_kernels.s
