import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")][0]
rows = c.execute(f"select start, end, size, name_id, src_agent_id, dst_agent_id, stream_id from {mc} order by start").fetchall()
print(len(rows), "copies")
h = collections.Counter((r[2], r[3], r[4], r[5]) for r in rows)
for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:25]:
    print(v, k)
t_end = rows[-1][1]
# the h2d leg: find 2D-copy sized entries
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kr = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
names = collections.Counter(n.split("(")[0][:60] for n, a, b in kr)
for k, v in names.most_common(30): print(v, k)
# timeline around the middle of k_detect launches
det = [(a, b) for n, a, b in kr if "k_detect" in n]
mid = det[len(det) * 3 // 4][0]
print("--- events in 4 ms after detect #", len(det) * 3 // 4)
ev = [(a, b, "K " + n.split("(")[0].replace("orbx::", "")[:40]) for n, a, b in kr if mid <= a < mid + 4e6 and ("k_detect" in n or "copy" in n.lower() or "k_describe" in n)]
ev += [(r[0], r[1], "C size=%d name=%s src=%s dst=%s st=%s" % (r[2], r[3], r[4], r[5], r[6])) for r in rows if mid <= r[0] < mid + 4e6]
for a, b, s in sorted(ev)[:90]:
    print("%9.1f %8.1f  %s" % ((a - mid) / 1e3, (b - a) / 1e3, s))
