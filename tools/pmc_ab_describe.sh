#!/bin/bash
# PMC counters of k_describe for several builds on ONE box: tools/pmc_ab_describe.sh liborbx_a.so liborbx_b.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" \
             "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    rm -rf /tmp/pm_$i
    ORBX_LIB_NAME=$L KB_NOPROF=1 timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pm_$i -o pmc -- python $R/tools/kbench.py 32 3 > /dev/null 2> /tmp/pm_$i.err
    db=$(find /tmp/pm_$i -name "*.db" | head -1)
    if [ -n "$db" ]; then echo "== $L group $i"; python $R/tools/pmc_insts.py $db k_describe /tmp/pm_$i.json; else echo "$L group $i: no output: $(tail -2 /tmp/pm_$i.err)"; fi
  done
done
