#!/bin/bash
# Parity + throughput + per-kernel durations of the pre-processing chain (run through gpurun from the repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-rX}
cd $R
python -m pytest tests/test_cpp_mirror.py tests/test_rectify_clahe.py tests/test_preprocess.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_preproc.py > $O/${TAG}_preproc.json 2>/dev/null; cat $O/${TAG}_preproc.json
cd /tmp && export TMPDIR=/tmp
for what in rectify clahe; do
  rocprofv3 --kernel-trace --stats -d $O/pp_$what -o pp -- python $R/tools/bench_preproc.py $what > /dev/null 2>&1
  cd $R; python tools/rocprof_summary.py $(find $O/pp_$what -name "*.db" | head -1) $O/${TAG}_preproc_${what}_stats.csv > /dev/null
  sed 's/(orbx::[A-Za-z]*Args[^"]*)//' $O/${TAG}_preproc_${what}_stats.csv | cut -c1-90; rm -rf $O/pp_$what; cd /tmp
done
