#!/bin/bash
# k_resize tile order A/B on the GPU box: per-level durations (kernel trace, one handle) and FETCH / WRITE bytes per launch
#   tools/resize_ab.sh "1 4 8" > gpurun_out/resize_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for run in ${1:-1 8}; do
  D=$(mktemp -d /tmp/rsab.XXXXXX)
  echo "== ORBX_RESIZE_XCD_RUN=$run"
  ORBX_RESIZE_XCD_RUN=$run rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- python $R/bench.py --no-extras --cpu-pairs 0 --handles 1 --steps 30 --warmup 3 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one handle: %.4f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))"
  python $R/tools/rocprof_levels.py $(find $D/kt -name "*.db" | head -1) k_resize
  for c in FETCH_SIZE WRITE_SIZE; do
    ORBX_RESIZE_XCD_RUN=$run rocprofv3 --pmc $c --kernel-trace -d $D/pmc_$c -o pmc -- python $R/bench.py --no-extras --cpu-pairs 0 --handles 1 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $(find $D/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $D/pmc_WRITE_SIZE -name "*.db" | head -1) $D/traffic.json ab > /dev/null 2>&1
  python -c "import json; d=json.load(open('$D/traffic.json')); [print(n, v) for n, v in d.items() if 'resize' in n]"
  ORBX_RESIZE_XCD_RUN=$run python $R/bench.py --no-extras --cpu-pairs 0 --steps 100 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three handles: %.4f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))"
  rm -rf $D
done
