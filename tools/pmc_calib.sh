#!/bin/bash
# Counter calibration on the GPU box (run through gpurun from the repo root): tools/pmc_calib.sh <tag>
#  1. FETCH_SIZE / WRITE_SIZE against known byte counts per access shape   -> gpurun_out/<tag>_pmc_calibration.json
#  2. SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES on saturated loops of ONE instruction class each
#     (2-cycle, 4-cycle and 8-cycle classes of tools/ubench/valu_issue) -> gpurun_out/<tag>_pmc_valu_calibration.txt
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/traffic_calib > /tmp/known.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -o pmc -- $R/tools/ubench/traffic_calib > /dev/null 2> /tmp/cal_$c.err
done
python $R/tools/pmc_calib.py /tmp/known.json $(find /tmp/cal_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/cal_WRITE_SIZE -name "*.db" | head -1) \
  $O/${TAG}_pmc_calibration.json > $O/${TAG}_pmc_calibration.txt 2>&1
CLASSES=add_u32,and_b32,lshrrev_b32,min_u16,fma_f32,lshlrev_b32,perm_b32,add_u32_sdwa,pk_maximum3_f16,dot4_u32_u8,cmp_gt_u32,add_u32_sgpr,exp_f32
: > $O/${TAG}_pmc_valu_calibration.txt
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAVES SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/cal_v
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/cal_v -o pmc -- $R/tools/ubench/valu_issue --sat --only $CLASSES > /tmp/cal_v.out 2> /tmp/cal_v.err
  db=$(find /tmp/cal_v -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_insts.py $db k_ >> $O/${TAG}_pmc_valu_calibration.txt; else echo "group ($grp): no output: $(tail -3 /tmp/cal_v.err)" >> $O/${TAG}_pmc_valu_calibration.txt; fi
done
cat /tmp/cal_v.out >> $O/${TAG}_pmc_valu_calibration.txt
cat $O/${TAG}_pmc_calibration.txt $O/${TAG}_pmc_valu_calibration.txt
