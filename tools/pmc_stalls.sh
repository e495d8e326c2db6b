#!/bin/bash
# Stall / LDS-conflict counters of the pipeline's kernels (run through gpurun from the repo root):
#   tools/pmc_stalls.sh <tag>      -> gpurun_out/<tag>_pmc_stalls.txt
# One rocprofv3 --pmc pass per counter group (kernel trace only, no other trace domains).
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_pmc_stalls.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/${TAG}_ps_$i -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile > /dev/null 2> $O/${TAG}_ps_$i.err
  db=$(find $O/${TAG}_ps_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_insts.py $db k_ /tmp/ps_$i.json >> $O/${TAG}_pmc_stalls.txt; else echo "group $i ($grp): no output: $(tail -2 $O/${TAG}_ps_$i.err)" >> $O/${TAG}_pmc_stalls.txt; fi
  rm -rf $O/${TAG}_ps_$i $O/${TAG}_ps_$i.err
done
cat $O/${TAG}_pmc_stalls.txt
