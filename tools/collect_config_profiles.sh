#!/bin/bash
# rocprofv3 evidence for the BASELINE configurations that are not the headline:
#   tools/collect_config_profiles.sh r4      (on the GPU box, from the repo root, through gpurun)
# per configuration: --kernel-trace --stats of a one-handle run (un-overlapped kernel durations) and the separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; instruction counters) -> gpurun_out/<tag>_<cfg>_{kernel_stats_1handle.csv,pmc_traffic.json,pmc_insts.txt}
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_cfg() {
  local cfg=$1; shift
  local args="$@"
  rm -rf /tmp/cp_kt /tmp/cp_f /tmp/cp_w /tmp/cp_i
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cp_kt -o kt -- python $R/bench.py --no-extras --handles 1 $args > $O/${TAG}_${cfg}_bench_1handle_under_rocprof.json 2> /dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/cp_f -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile $args > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/cp_w -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile $args > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/cp_i -o pmc -- python $R/bench.py --no-extras --handles 1 --steps 5 --warmup 2 --no-profile $args > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/cp_kt -name "*.db" | head -1) $O/${TAG}_${cfg}_kernel_stats_1handle.csv > /dev/null
  python $R/tools/pmc_summary.py $(find /tmp/cp_f -name "*.db" | head -1) $(find /tmp/cp_w -name "*.db" | head -1) $O/${TAG}_${cfg}_pmc_traffic.json ${TAG}_${cfg} > /dev/null
  python $R/tools/pmc_insts.py $(find /tmp/cp_i -name "*.db" | head -1) k_ > $O/${TAG}_${cfg}_pmc_insts.txt
  echo "== $cfg"; head -12 $O/${TAG}_${cfg}_kernel_stats_1handle.csv
}
ONLY=${2:-all}   # second argument: one configuration (C2 | S640 | C4 | C5) instead of all four
want() { [ "$ONLY" = all ] || [ "$ONLY" = "$1" ]; }
want C2 && run_cfg C2 --mode mono --width 640 --height 480 --nfeatures 1000
want S640 && run_cfg S640 --mode stereo --width 640 --height 480 --nfeatures 1000
want C4 && run_cfg C4 --mode fisheye --width 512 --height 512 --nfeatures 1500
want C5 && run_cfg C5 --config C5
true
