"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel launch count / total / average duration, and
PMC counter sums per kernel when present.
usage: python tools/rocpd_summary.py <results.db> [--json out.json]   (--json: per-kernel per-call counter values)"""
import json
import re
import sqlite3
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('dz::', '')
    name = re.sub(r'TileCfg<(\d+), (\d+), (\d+), \d+, \d+>', r'\1x\2x\3', name)
    name = re.sub(r'<HTile<(\d+), (\d+), (\d+), \d+, \d+>, Math(\w+?)(, \w+)*>', r'<\1x\2x\3>[\4]', name)
    # resident-tile 3x3 kernel: <BC, Math, OUT_F32, threads, diag, rows per wave> -> <tile rows x 32 x BC>[Math] (16-row tiles at BC = 32)
    # (round 5: a trailing `true` = the sparse-input form, dz_conv2d_desc.in_rowidx -> "|rows")
    name = re.sub(r'k_conv3x3_h<(\d+), Math(\w+?), \w+((?:, \d+)*)(, (?:true|false))?>',
                  lambda m: 'k_conv3x3_h<%sx32x%s%s>[%s]' % ('16' if m.group(1) == '32' else '8', m.group(1), '|rows' if (m.group(4) or '').endswith('true') else '',
                                                          m.group(2)), name)
    return name[:90]


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += (e - s)
    total = sum(a[1] for a in agg.values())
    print('# kernel-trace summary of %s : %d dispatches, %.3f ms total device time' % (db.split('/')[-2] if '/' in db else db, len(rows), total / 1e6))
    print('%-92s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-92s %8d %12.1f %10.2f %6.2f' % (k, n, t / 1e3, t / 1e3 / n, 100.0 * t / total))
    try:
        pm = cur.execute("select * from counters_collection limit 1").fetchall()
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if pm:
            ni, ci, vi = ccols.index('kernel_name') if 'kernel_name' in ccols else ccols.index('name'), ccols.index('counter_name'), ccols.index('value')
            agg2 = {}
            for r in cur.execute("select * from counters_collection"):
                key = (short(r[ni]), r[ci])
                a = agg2.setdefault(key, [0, 0.0])
                a[0] += 1
                a[1] += float(r[vi])
            print('\n# PMC counters (sum over dispatches / per dispatch)')
            for (k, c), (n, v) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:60]:
                print('%-80s %-14s calls %6d sum %16.1f per_call %14.1f' % (k, c, n, v, v / n))
            if '--json' in sys.argv:
                out = {}
                for (k, c), (n, v) in agg2.items():
                    out.setdefault(k.strip(), {})[c] = {'calls': n, 'per_call': v / n}
                with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
                    json.dump(out, f, indent=1, sort_keys=True)
    except Exception as ex:  # noqa
        print('no counters:', ex)


if __name__ == '__main__':
    main()
