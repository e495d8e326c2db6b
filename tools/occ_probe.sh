#!/bin/bash
# average resident waves of the pipeline's kernels: SQ_LEVEL_WAVES / SQ_BUSY_CYCLES style counters (one pmc pass each)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for set in "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACCUM_PREV_HIRES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/occ
  KB_NOPROF=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/occ -o pmc -- python $R/tools/kbench.py 32 3 > /dev/null 2> /tmp/occ.err
  db=$(find /tmp/occ -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_insts.py $db k_ | grep "k_detect\|k_describe\|k_resize "; else echo "no output for [$set]: $(tail -2 /tmp/occ.err)"; fi
done
