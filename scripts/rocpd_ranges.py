"""The library's ROCTx ranges (VD_ROCTX=1; csrc/common.h VdRange: one per step phase, host-side enqueue ranges) as recorded by
`rocprofv3 --marker-trace` in a rocpd database (view `regions`, category MARKER_CORE_RANGE_API, the range name in extdata.message):
name, count, mean duration.      python scripts/rocpd_ranges.py <db>"""
import json
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    print("# ROCTx ranges recorded by rocprofv3 --marker-trace (VD_ROCTX=1): range, count, mean ms -- HOST-side enqueue ranges of the step phases")
    print("# (the kernels dispatched inside a range carry its correlation in the same database; the device time of a phase is in the kernel stats)")
    agg = defaultdict(list)
    try:
        rows = db.execute("select name, start, end, extdata from regions where category like 'MARKER%'").fetchall()
    except sqlite3.Error as exc:
        print("# no `regions` view in this database:", exc)
        return
    for name, start, end, ext in rows:
        try:
            msg = json.loads(ext).get('message') or name
        except Exception:
            msg = name
        agg[msg].append((end - start) / 1e6)
    for msg, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-52s %6d %10.3f" % (msg, len(v), sum(v) / len(v)))
    if not agg:
        print("# no marker ranges (was VD_ROCTX=1 set, and --marker-trace given?)")


if __name__ == '__main__':
    main(sys.argv[1])
