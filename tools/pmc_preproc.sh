#!/bin/bash
# PMC counters of the pre-processing kernels (run through gpurun from the repo root): tools/pmc_preproc.sh <tag> [lib]
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_preproc_pmc.txt
for what in rectify clahe; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rm -rf /tmp/pp_$i
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pp_$i -o pmc -- python $R/tools/bench_preproc.py $what > /dev/null 2> /tmp/pp_$i.err
    db=$(find /tmp/pp_$i -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/pmc_insts.py $db k_ >> $O/${TAG}_preproc_pmc.txt; else echo "$what group $i ($grp): no output: $(tail -1 /tmp/pp_$i.err | cut -c1-200)" >> $O/${TAG}_preproc_pmc.txt; fi
  done
done
python - "$O/${TAG}_preproc_pmc.txt" "$O/${TAG}_preproc_traffic.json" <<'PY'
import json, re, sys
# HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 tallies 128-byte read requests at 64 bytes: tools/pmc_summary.py)
f, w = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(k_\w+) (FETCH_SIZE|WRITE_SIZE)=(\d+)", line)
    if m:
        (f if m.group(2) == "FETCH_SIZE" else w)[m.group(1)] = int(m.group(3))
out = {k: int((2 * f.get(k, 0) + w.get(k, 0)) * 1024) for k in set(f) | set(w)}
out["_note"] = "HBM-side bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024, separate rocprofv3 --pmc passes over tools/bench_preproc.py (64 frames per launch)"
out["_raw_KiB"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
cat $O/${TAG}_preproc_pmc.txt
