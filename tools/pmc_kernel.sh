#!/bin/bash
# SQ / traffic counters of ONE kernel (substring match) of any command, separate --pmc passes (--kernel-trace only):
#   tools/pmc_kernel.sh <tag> <kernel-substring> <command ...>      -> gpurun_out/pmc_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KSUB=$2; shift; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o p$i -- "$@" > $OUT/run$i.log 2>&1)
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel ~ $KSUB: per-dispatch averages")
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):18.1f}   (n={len(v)})")
PY
cat $OUT/summary.txt
