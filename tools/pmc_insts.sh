#!/bin/bash
# dynamic instruction mix of the VFO-bank kernels (rocprofv3 PMC, own run without --stats): vector / matrix / LDS / scalar instructions per launch
set -u
O=gpurun_out/pmc_insts
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
cd /tmp; export TMPDIR=/tmp
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $ctr | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $R/$O/$tag -o p -- python $R/tools/vfo_only_time.py 16777216 32 4 > $R/$O/$tag.log 2>&1
    db=$(find $R/$O/$tag -name "*.db" | head -1)
    python $R/tools/rocpd_summary.py $db --pmc $db --out $R/$O/$tag.md --title "$ctr" 2>&1 | tail -1
    grep -E "vfo_pipe|vfo_frontcm" $R/$O/$tag.md | grep -v "^| vfo.*| [0-9]* | [0-9.]* | [0-9.]* |" | head -8
    find $R/$O/$tag -name "*.db" -size +4M -delete
done
