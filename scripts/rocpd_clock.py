"""Effective shader clock and MFMA-busy per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES run
(rocpd .db): clock = GRBM_GUI_ACTIVE / dispatch duration; busy = SQ_VALU_MFMA_BUSY_CYCLES * 32 / (1024 * GRBM_GUI_ACTIVE).
Splits the run into quarters of the dispatch sequence to show drift under sustained load."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'void (gemm_f32\w*)_kernel<GemmCfg<([\d, ]+)>, (.*)>$', name)
    return ('%s<%s|%s>' % (m.group(1), m.group(2).replace(', ', 'x'), m.group(3)))[:70] if m else name.replace('void ', '')[:70]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events").fetchall()
    disp = defaultdict(dict)
    for name, did, dur, cn, v in rows:
        disp[did]['name'], disp[did]['dur'] = short(name), dur
        disp[did][cn] = disp[did].get(cn, 0) + v
    ids = sorted(disp)
    q = max(1, len(ids) // 4)
    agg = defaultdict(lambda: [[0, 0, 0, 0] for _ in range(4)])
    for i, did in enumerate(ids):
        d = disp[did]
        if 'GRBM_GUI_ACTIVE' not in d or d['dur'] <= 0:
            continue
        a = agg[d['name']][min(3, i // q)]
        a[0] += d['GRBM_GUI_ACTIVE']; a[1] += d['dur']; a[2] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0); a[3] += 1
    print("%-72s %s" % ("kernel", "  ".join("q%d: n  GHz  busy%%  avg_us" % (k + 1) for k in range(4))))
    for name, qs in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:12]:
        cells = []
        for g, dur, mf, n in qs:
            cells.append("%4d %.2f %5.1f %8.1f" % (n, g / dur if dur else 0, 100.0 * mf * 32 / (1024.0 * g) if g else 0,
                                                   dur / n / 1e3 if n else 0))
        print("%-72s %s" % (name, "   ".join(cells)))


if __name__ == '__main__':
    main(sys.argv[1])
