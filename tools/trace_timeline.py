"""Timeline of ONE step from a rocprofv3 rocpd database: kernels in start order with start offset, duration and
the idle gap before each one (development tool).
usage: python tools/trace_timeline.py <results.db> [step_index_from_end] [first-kernel-of-a-step]
A step ends with k_pack_detections (the detector) unless the name of the FIRST kernel of a step is given (e.g. k_level_keys)."""
import re
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
con = sqlite3.connect(db)
rows = con.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
first = sys.argv[3] if len(sys.argv) > 3 else None
marks = [i for i, r in enumerate(rows) if (first or 'k_pack_detections') in r[0]]
a, b = (marks[-back - 1], marks[-back]) if first else (marks[-back - 1] + 1, marks[-back] + 1)
t0 = rows[a][1]
busy_end = t0
tot_gap = 0.0
agg = {}
for r in rows[a:b]:
    n = re.sub(r'\(.*', '', r[0]).replace('void ', '').replace('dz::', '')[:58]
    gap = max(0.0, (r[1] - busy_end) / 1e3)
    tot_gap += gap
    busy_end = max(busy_end, r[2])
    print('%9.1f %8.1f us gap %6.1f grid %7d  %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3] // max(r[4], 1), n))
print('step wall %.1f us, idle gaps %.1f us, kernels %d' % ((busy_end - t0) / 1e3, tot_gap, b - a))
