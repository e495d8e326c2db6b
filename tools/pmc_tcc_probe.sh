cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r3_counters_list.txt 2>&1
grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_HIT[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*\|TCP_TCC_READ[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" $R/gpurun_out/r3_counters_list.txt | sort -u | head -60
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/cal_t
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/cal_t -o pmc -- $R/tools/ubench/traffic_calib > /dev/null 2> /tmp/cal_t.err
  db=$(find /tmp/cal_t -name "*.db" | head -1)
  if [ -n "$db" ]; then python - $db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
key = "dispatch_id" if "dispatch_id" in cols else "rowid"
rows = c.execute("select %s, kernel_name, counter_name, sum(value) from counters_collection where kernel_name like '%%k_%%' group by 1,3 order by 1,3" % key).fetchall()
for r in rows: print(r[0], r[1].split("(")[0], r[2], int(r[3]))
PY
  else echo "group $grp failed: $(tail -3 /tmp/cal_t.err)"; fi
done
