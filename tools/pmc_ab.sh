#!/bin/bash
# A/B counters of k_detect variants on the GPU box: tools/pmc_ab.sh <tag> [env assignments ...]
# e.g. tools/pmc_ab.sh nc2 ORBX_DETECT_NC=2   -> gpurun_out/pmcab_<tag>.txt (per-wave instruction / cycle / stall counters)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmcab_$TAG.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmcab_$i
  env "$@" KB_NOPROF=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcab_$i -o pmc -- python $R/tools/kbench.py 32 3 > /dev/null 2>&1
  python $R/tools/pmc_insts.py $(find /tmp/pmcab_$i -name "*.db" | head -1) k_detect >> $O/pmcab_$TAG.txt
done
cat $O/pmcab_$TAG.txt
