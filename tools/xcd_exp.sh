#!/bin/bash
# XCD-run experiment: HBM-side fetch and duration of k_detect / k_blur against the run length of the block->tile remap.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
for K in "8 1" "8 2" "8 4" "8 8" "16 10" "32 20"; do
  set -- $K
  export ORBX_DETECT_XCD_RUN=$1 ORBX_BLUR_XCD_RUN=$2
  cd $R
  echo "detect run=$1 blur run=$2"
  python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "extract" 2>&1 | tail -1
  python bench.py --cpu-pairs 0 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.readline()); r=b['roofline']
print('pairs/s', b['value'], 'ms', b['ms_per_step'], 'detect us', r['avg_launch_us'], r['isolated_avg_launch_us'], 'stages', {k: round(v['avg_us'],1) for k,v in b.get('stages',{}).items()} if isinstance(b.get('stages'),dict) else '')"
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/x_$1_$2 -o pmc -- python $R/bench.py --cpu-pairs 0 --handles 1 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
  cd $R
  db=$(find $O/x_$1_$2 -name "*.db" | head -1)
  python tools/pmc_summary.py $db $db /tmp/x.json x > /dev/null 2>&1
  python -c "import json; d=json.load(open('/tmp/x.json')); print('fetch MB: detect %.0f blur %.0f describe %.0f' % tuple(d[k]['fetch_KiB']*1024/1e6 for k in ('k_detect','k_blur','k_describe')))"
  python tools/rocprof_summary.py $db /tmp/x.csv > /dev/null 2>&1; grep -E "k_blur|k_detect" /tmp/x.csv | sed 's/(orbx::Geom[^"]*"//' | cut -c1-80
  rm -rf $O/x_$1_$2
done
