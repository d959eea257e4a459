#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel stats (like --stats) and, when present, PMC counters
averaged per dispatch.  usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels").fetchall()
st = collections.defaultdict(list)
for n, s, e in rows:
    st[n.split('(')[0]].append((e - s) / 1e3)
tot = sum(sum(v) for v in st.values())
out.write('%-62s %8s %12s %10s %10s %10s %6s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    out.write('%-62s %8d %12.1f %10.2f %10.2f %10.2f %6.2f\n' % (n[:62], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
try:
    pm = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
except Exception as e:
    try:
        ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        out.write('counters_collection columns: %s\n' % ccols)
        pm = []
    except Exception:
        pm = []
if pm:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v in pm:
        acc[k.split('(')[0]][c].append(v)
    out.write('\nPMC counters (average per dispatch)\n')
    for k, d in sorted(acc.items()):
        out.write(k[:70] + '\n')
        for c, v in sorted(d.items()):
            out.write('     %-28s %16.1f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
