#!/usr/bin/env python3
"""Timeline of one MD step from a rocprofv3 rocpd database: kernels between two consecutive belief-propagation solves
(one per force pass) with start offset, duration, queue and grid.  usage: rocpd_timeline.py results.db [step_index] [out.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ('name', 'start', 'end', 'queue_id', 'stream_id', 'grid_x', 'grid_y', 'workgroup_x', 'lds_size') if c in cols]
rows = db.execute("select %s from kernels order by start" % ','.join(want)).fetchall()
marks = [i for i, r in enumerate(rows) if 'k_rotamer_bp<' in r[0] or 'k_rotamer_bp_cluster<true>' in r[0]]
marks = [m for k, m in enumerate(marks) if k == 0 or m - marks[k - 1] > 3]      # (a cluster solve is followed by its fallback launch)
a, b = marks[idx], marks[idx + 1]
t0 = rows[a][2]
out.write('columns: %s\n' % want)
prev_end = t0
for r in rows[a + 1:b + 1]:
    d = dict(zip(want, r))
    out.write('%9.1f us  +%8.1f us  gap %7.1f  q%-3s %-40s grid(%s,%s) wg %s lds %s\n' % (
        (d['start'] - t0) / 1e3, (d['end'] - d['start']) / 1e3, (d['start'] - prev_end) / 1e3,
        d.get('queue_id', d.get('stream_id', '?')), d['name'].split('(')[0][:40], d.get('grid_x'), d.get('grid_y'), d.get('workgroup_x'), d.get('lds_size')))
    prev_end = max(prev_end, d['end'])
out.write('step wall %.1f us\n' % ((rows[b][2] - t0) / 1e3))
