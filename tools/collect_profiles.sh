#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r1d
# writes gpurun_out/<tag>_* ; copy what should be judged into profiles/ afterwards (see profiles/README.md).
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o kt -- python $R/bench.py --no-extras --handles ${HANDLES:-3} > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
# second kernel trace with ONE handle: un-overlapped kernel durations (the counters below are per launch either way)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt1 -o kt -- python $R/bench.py --no-extras --handles 1 > $O/${TAG}_bench_1handle_under_rocprof.json 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_pmc_$c -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/${TAG}_pmc_insts -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/${TAG}_kt -name "*.db" | head -1) $O/${TAG}_kernel_stats.csv
python tools/rocprof_summary.py $(find $O/${TAG}_kt1 -name "*.db" | head -1) $O/${TAG}_kernel_stats_1handle.csv
python tools/pmc_summary.py $(find $O/${TAG}_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/${TAG}_pmc_WRITE_SIZE -name "*.db" | head -1) $O/${TAG}_pmc_traffic.json $TAG > /dev/null
python tools/pmc_insts.py $(find $O/${TAG}_pmc_insts -name "*.db" | head -1) k_ $O/${TAG}_pmc_insts.json > $O/${TAG}_pmc_insts.txt
python tools/rocprof_levels.py $(find $O/${TAG}_kt1 -name "*.db" | head -1) k_resize > $O/${TAG}_resize_levels.txt
rm -rf $O/${TAG}_kt $O/${TAG}_kt1 $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $O/${TAG}_pmc_insts
# the secondary evidence profiles/README.md lists (skip with QUICK=1)
if [ -z "${QUICK:-}" ]; then
  bash tools/bench_configs.sh > $O/${TAG}_configs.txt 2>&1
  python bench.py --config C5 --no-extras 2> /dev/null | tail -1 > $O/${TAG}_bench_c5.json
  python tools/latency.py > $O/${TAG}_latency.txt 2>&1
  python bench.py --steps 20 --warmup 3 2> /dev/null | tail -1 > $O/${TAG}_bench_20steps.json
  for h in 1 2 4; do python bench.py --no-extras --handles $h 2> /dev/null | tail -1 > $O/${TAG}_bench_h$h.json; done
  python tools/cell_stats.py 8 > $O/${TAG}_cell_stats.txt 2>&1
  bash tools/pmc_stalls.sh $TAG > /dev/null 2>&1
fi
cat $O/${TAG}_bench.json | head -c 600; echo; cat $O/${TAG}_kernel_stats.csv; cat $O/${TAG}_kernel_stats_1handle.csv; cat $O/${TAG}_pmc_insts.txt
