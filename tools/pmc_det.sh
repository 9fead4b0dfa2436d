#!/bin/bash
# Runs on the GPU box: two rocprofv3 --pmc passes (SQ counters only, kept apart from any trace) over the det-only layer profile,
# then per-kernel averages of the kernels whose name contains $1 (default k_fpn).  Output under gpurun_out/$2.
sub=${1:-k_fpn}
out=gpurun_out/${2:-pmc_det}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/a -o a -- python3 tools/layer_profile.py 32 1 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $out/b -o b -- python3 tools/layer_profile.py 32 1 0 > /dev/null 2>&1
python3 tools/scratch/pmc_kernel.py "$sub" $(find $out -name "*counter_collection.csv") > $out/summary.txt 2>&1
cat $out/summary.txt
