#!/bin/bash
# Per-kernel average durations of one bench.py configuration from a rocprofv3 kernel trace (run on the GPU box):
#   tools/ktrace.sh [bench.py args...]     e.g.  tools/ktrace.sh --handles 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d /tmp/ktrace.XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --no-extras --cpu-pairs 0 --steps 30 --warmup 3 "$@" > $D/bench.json 2> /dev/null
cd $R
python tools/rocprof_summary.py $(find $D -name "*.db" | head -1) $D/stats.csv > /dev/null
python - $D/stats.csv <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
    if r[0].startswith("__amd") or "at::native" in r[0] or "clock_probe" in r[0]: continue
    name = r[0].split("(")[0].replace("void ", "").replace("orbx::", "")
    print("%-28s calls %5s  avg %8.2f us" % (name[:28], r[1], float(r[3])))
PY
python -c "import json,sys; d=json.loads(open('$D/bench.json').read().strip().splitlines()[-1]); print('step %.4f ms  %.0f pairs/s' % (d['ms_per_step'], d['value']))"
rm -rf $D
