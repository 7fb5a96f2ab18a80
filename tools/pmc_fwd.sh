#!/bin/bash
# PMC collection for the forward scan kernel (separate passes; --kernel-trace only, as gpurun requires)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for i in 1 2 3 4 5; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT";;
    2) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM";;
    3) C="FETCH_SIZE";;
    4) C="WRITE_SIZE";;
    5) C="GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA";;
  esac
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o p$i -- python $R/tools/prof_fwd.py > $OUT/run$i.log 2>&1
done
ls $OUT
