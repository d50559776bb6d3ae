"""Compact per-stream timeline of the last training step in a rocpd kernel trace."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'void gemm_f32_kernel<GemmCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>, (\w+), (\w+), (\w+)', name)
    if m:
        g = m.groups()
        return 'gemm<%sx%sx%sx%s%s,%s,%s,%s>' % (g[0], g[1], g[2], g[3], 'D' if g[4] == '1' else 'S', g[6], g[7], g[8])
    return name.replace('void ', '')[:50]


def main(path, nsteps_back=1):
    db = sqlite3.connect(path)
    rows = db.execute("select name, stream_id, start, end from kernels order by start").fetchall()
    # step boundary = clamp_adam_kernel
    adam = [i for i, r in enumerate(rows) if 'clamp_adam' in r[0]]
    lo, hi = adam[-1 - nsteps_back] + 1, adam[-1] + 1
    step = rows[lo:hi]
    t0 = step[0][2]
    print("step span %.3f ms, %d kernels" % ((step[-1][3] - t0) / 1e6, len(step)))
    try:    # which hardware (AQL) queue carried each stream's dispatches
        qs = db.execute("select stream_id, queue_id, count(*) from kernels group by stream_id, queue_id").fetchall()
        print("stream -> hardware queue (dispatches): " + ", ".join("s%s->q%s (%d)" % q for q in qs))
    except Exception as exc:
        print("(no queue ids in this trace: %s)" % exc)
    streams = sorted(set(r[1] for r in step))
    for s in streams:
        ks = [r for r in step if r[1] == s]
        busy = sum(r[3] - r[2] for r in ks)
        print("== stream %s: %d kernels, busy %.3f ms, from %.3f to %.3f ms" % (s, len(ks), busy / 1e6, (ks[0][2] - t0) / 1e6, (ks[-1][3] - t0) / 1e6))
        # run-length groups
        i = 0
        while i < len(ks):
            j = i
            nm = short(ks[i][0])
            while j + 1 < len(ks) and short(ks[j + 1][0]) == nm:
                j += 1
            dur = sum(r[3] - r[2] for r in ks[i:j + 1])
            print("   %8.3f -> %8.3f ms  x%-3d busy %7.3f ms  %s" % ((ks[i][2] - t0) / 1e6, (ks[j][3] - t0) / 1e6, j - i + 1, dur / 1e6, nm))
            i = j + 1


if __name__ == '__main__':
    main(sys.argv[1])
