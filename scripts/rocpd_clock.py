"""Shader clock per kernel from ONE rocprofv3 run with `--kernel-trace --pmc GRBM_GUI_ACTIVE`: average GRBM_GUI_ACTIVE cycles of a
kernel / its average duration (the counter run serialises dispatches, so durations are stand-alone durations)."""
import sqlite3
import sys
from collections import defaultdict

from rocpd_pmc import short


def main(path):
    db = sqlite3.connect(path)
    dur = defaultdict(list)
    for name, d in db.execute("select name, duration from kernels"):
        dur[short(name)].append(d)
    cyc = defaultdict(list)
    for name, cn, v in db.execute("select name, counter_name, counter_value from pmc_events"):
        if cn == 'GRBM_GUI_ACTIVE':
            cyc[short(name)].append(v)
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if k not in cyc or len(dur[k]) < 8:
            continue
        d = sorted(dur[k])[len(dur[k]) // 2] * 1e-3      # median, us
        c = sorted(cyc[k])[len(cyc[k]) // 2]
        print("%-84s n=%4d  median %9.1f us  GRBM_GUI_ACTIVE %11.0f  -> %.3f GHz" % (k, len(dur[k]), d, c, c / d * 1e-3))


if __name__ == '__main__':
    main(sys.argv[1])
