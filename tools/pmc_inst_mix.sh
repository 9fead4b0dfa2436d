#!/bin/bash
# Runs on the GPU box: dynamic instruction mix (VALU / scalar / LDS instructions per MFMA) of every kernel of one det pass over 32
# pages, from SQ_INSTS_* counters (own pass, no trace domains).  An fp32 MFMA loop pays its VALU instructions in MFMA time
# (DESIGN.md 5.4): this is where to look first.  Usage: gpurun -- 'bash tools/pmc_inst_mix.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/det_insts
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/det_insts -o s -- python3 tools/layer_profile.py 32 1 0 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/det_insts/**/s_counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_INSTS_VALU": n[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("GRBM_GUI_ACTIVE",0))[:22]:
    mf=v.get("SQ_INSTS_MFMA",0) or 1
    print("%-60s n=%3d valu/mfma %.2f salu/mfma %.2f lds/mfma %.2f mfma_busy(raw ratio) %.3f cycles/launch %.0f" % (k,n[k],(v["SQ_INSTS_VALU"]-v.get("SQ_INSTS_MFMA",0))/mf, v["SQ_INSTS_SALU"]/mf, v["SQ_INSTS_LDS"]/mf, v["SQ_VALU_MFMA_BUSY_CYCLES"]/max(v["GRBM_GUI_ACTIVE"],1)/(256*4), v["GRBM_GUI_ACTIVE"]/max(n[k],1)))
PY
