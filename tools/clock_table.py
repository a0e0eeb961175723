"""Idle CUs or clock?  From ONE rocprofv3 pass with --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES (+ --kernel-trace): per kernel
  effective clock  = GRBM_GUI_ACTIVE per launch / launch duration          (MI355X_MICROARCH.md, "DVFS give-back")
  busy-CU fraction = SQ_BUSY_CU_CYCLES per launch / (GRBM_GUI_ACTIVE per launch x CUs)   (share of the elapsed CU-cycles in which a CU held a wave)
The counters come summed over the 8 XCDs of the MI355X: GRBM_GUI_ACTIVE is divided by the number of XCDs (--xcds, default 8) to get
elapsed cycles; SQ_BUSY_CU_CYCLES is in quad-cycles of a CU's sequencer when --quad (default: checked against the ceiling and said).
usage: python tools/clock_table.py <results.db> [--cus 256] [--xcds 8]"""
import re
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_summary import short  # noqa: E402


def main():
    db = sys.argv[1]
    arg = lambda k, d: float(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d      # noqa: E731
    cus, xcds = arg('--cus', 256.0), arg('--xcds', 8.0)
    cur = sqlite3.connect(db).cursor()
    dur = {}
    for name, s, e in cur.execute("select name, start, end from kernels"):
        a = dur.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ni = ccols.index('kernel_name') if 'kernel_name' in ccols else ccols.index('name')
    ci, vi = ccols.index('counter_name'), ccols.index('value')
    pmc = {}
    for r in cur.execute("select * from counters_collection"):
        a = pmc.setdefault((short(r[ni]), r[ci]), [0, 0.0])
        a[0] += 1
        a[1] += float(r[vi])
    print('# %s' % db)
    print('# effective clock = GRBM_GUI_ACTIVE / %d XCDs / launch time; busy CUs = SQ_BUSY_CU_CYCLES / (elapsed cycles x %d CUs)' % (xcds, cus))
    print('%-70s %6s %10s %12s %9s %9s' % ('kernel', 'calls', 'avg_us', 'GUI_ACTIVE', 'clock GHz', 'busy CUs'))
    rows = []
    for k, (n, t) in dur.items():
        g = pmc.get((k, 'GRBM_GUI_ACTIVE'))
        b = pmc.get((k, 'SQ_BUSY_CU_CYCLES'))
        if not g or t == 0:
            continue
        avg_ns = t / n
        cyc = g[1] / g[0] / xcds
        clock = cyc / avg_ns                   # cycles per ns = GHz
        busy = (b[1] / b[0]) / (cyc * cus) if b else float('nan')
        rows.append((t, k, n, avg_ns / 1e3, g[1] / g[0], clock, busy))
    for t, k, n, us, g, clock, busy in sorted(rows, reverse=True)[:24]:
        print('%-70s %6d %10.1f %12.0f %9.3f %9.3f' % (k[:70], n, us, g, clock, busy))


if __name__ == '__main__':
    main()
