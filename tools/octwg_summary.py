#!/usr/bin/env python3
"""Workgroup lives of k_octree from a -DOCT_WG_PROF run of tools/octree_frame_prof.py: per level, the mean over launches and images.
usage: python tools/octwg_summary.py <log>"""
import re, sys, collections
txt = open(sys.argv[1]).read()
w = [tuple(map(int, m.groups())) for m in re.finditer(r'octwg level (\d+) img (\d+) start (\d+) end (\d+)', txt)]
w.sort(key=lambda x: x[2])
groups = []
for x in w:
    if groups and x[2] - groups[-1][0][2] < 10000: groups[-1].append(x)
    else: groups.append([x])
groups = [g for g in groups if len(g) >= 16][1:]   # skip the first (cold) launch
per = collections.defaultdict(list); span = []
for g in groups:
    s0 = min(x[2] for x in g); span.append((max(x[3] for x in g) - s0) / 100)
    for x in g: per[x[0]].append((x[3] - x[2]) / 100)
print("%d launches: first start -> last end %.1f us (mean); life per level: %s" % (len(groups), sum(span) / len(span), "  ".join("L%d %.1f" % (l, sum(v) / len(v)) for l, v in sorted(per.items()))))
