"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / % (the
--stats table), optionally restricted to the last `--steps` training steps."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'void (gemm_f32\w*)_kernel<GemmCfg<([\d, ]+)>, (.*)>$', name)
    if m:
        return ('%s<%s|%s>' % (m.group(1), m.group(2).replace(', ', 'x'), m.group(3)))[:90]
    return name.replace('void ', '')[:90]


def main(path, training_steps_only=False):
    db = sqlite3.connect(path)
    where = ""
    if training_steps_only:
        # keep the dispatches up to the last optimiser launch: bench.py's `roofline.alone` leg (the dominant kernel run
        # by itself after the timed region) would otherwise be averaged into the in-step figures
        last = db.execute("select max(end) from kernels where name like '%clamp_adam%'").fetchone()[0]
        if last is not None:
            where = " where end <= %d" % last
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels" + where + " group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("%-78s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, c, t, a, mn, mx in rows:
        print("%-78s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short(n), c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                100.0 * t / total))
    span = db.execute("select min(start), max(end) from kernels" + where).fetchone()
    print("total kernel time %.3f ms over a %.3f ms span (%d dispatches)" % (
        total / 1e6, (span[1] - span[0]) / 1e6, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main(sys.argv[1], training_steps_only='--steps-only' in sys.argv[2:])
