"""Per-kernel averages of the PMC counters stored in a rocprofv3 rocpd .db (--pmc run)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'void gemm_f32_kernel<GemmCfg<(\d+), (\d+), (\d+), (\d+)>, (\w+), (\w+), (\w+)', name)
    if m:
        return 'gemm_f32<%sx%sx%sx%s,%s,%s,%s>' % m.groups()
    return name.replace('void ', '')[:80]


def main(path):
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('pmc_events')")]
    print("# pmc_events columns:", cols)
    q = "select name, counter_name, counter_value, dispatch_id from pmc_events" if 'counter_name' in cols else None
    if q is None:
        print(db.execute("select * from pmc_events limit 3").fetchall())
        return
    agg = defaultdict(lambda: defaultdict(list))
    for name, cn, v, did in db.execute(q):
        agg[short(name)][cn].append(v)
    for k, d in agg.items():
        n = max(len(v) for v in d.values())
        print("%s  (dispatches %d)" % (k, n))
        for cn, vals in sorted(d.items()):
            print("    %-32s avg %.4g" % (cn, sum(vals) / len(vals)))


if __name__ == '__main__':
    main(sys.argv[1])
