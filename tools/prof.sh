#!/bin/bash
# usage: tools/prof.sh <tag> <kernel-name-like-pattern (sqlite LIKE, e.g. "%abx%")> <python driver and args...>
# rocprofv3 kernel-trace stats + separate PMC passes (never combined with other trace domains);
# outputs under gpurun_out/prof_<tag>/ (summary.txt is what gets copied to profiles/).
# PMC_PASSES="A B C|D E" overrides the counter passes ('|' separates passes).
set -u
TAG=$1; shift
PAT=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT
DEFAULT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE|SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS|FETCH_SIZE|WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
IFS='|' read -r -a PASSES <<< "${PMC_PASSES:-$DEFAULT}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $ROOT/"$@" > $OUT/trace.log 2>&1
i=0
for PMC in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o p -- python $ROOT/"$@" > $OUT/pmc$i.log 2>&1
done
cd $ROOT
python tools/prof_summary.py $OUT "$PAT" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the rocpd databases are large (gpurun_out is capped at 64 MiB): keep the summary only unless PROF_KEEP_DB=1
if [ "${PROF_KEEP_DB:-0}" != "1" ]; then find $OUT -name "*.db" -delete; fi
