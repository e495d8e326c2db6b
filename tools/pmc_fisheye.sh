#!/bin/bash
# PMC counters of the fisheye association kernels (C4 workload, one handle) for several builds on ONE box:
#   tools/pmc_fisheye.sh liborbx.so [liborbx_x.so ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" \
             "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" \
             "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rm -rf /tmp/pm_$i
    ORBX_LIB_NAME=$L timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pm_$i -o pmc -- python $R/bench.py --no-extras --no-profile --handles 1 --steps 5 --warmup 2 --mode fisheye --width 512 --height 512 --nfeatures 1500 > /dev/null 2> /tmp/pm_$i.err
    db=$(find /tmp/pm_$i -name "*.db" | head -1)
    if [ -n "$db" ]; then echo "== $L group $i"; python $R/tools/pmc_insts.py $db k_fisheye; else echo "$L group $i: no output: $(tail -2 /tmp/pm_$i.err)"; fi
  done
done
