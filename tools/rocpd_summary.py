#!/usr/bin/env python
"""
Summarise rocprofv3 (rocpd sqlite) outputs into short text tables.

    python tools/rocpd_summary.py gpurun_out/prof_stats/r1_results.db            # kernel stats
    python tools/rocpd_summary.py --pmc gpurun_out/pmc_sq/r1_results.db [...]   # counters

Kernel names are shortened; torch's data-generation kernels are folded into one line.
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[0-9a-z, ]+>)?)', name)
    s = m.group(1) if m else name
    if s.startswith('at::native') or s.startswith('Cijk_'):
        return '[torch: synthetic-data generation] ' + s[:40]
    return s[:70]


def stats(db, steady=None):
    """Per-kernel table.  ``steady = (substring, n)``: a second line for the matching kernels over
    their LAST n launches only -- the iterations of the profiled command without its set-up
    launches (the fused PCA block times its plate pass on candidate allocations first)."""
    con = sqlite3.connect(db)
    rows = con.execute('select name, duration from kernels order by start').fetchall()
    agg = defaultdict(list)
    for n, d in rows:
        agg[short(n)].append(d)
    if steady:
        for k in [k for k in agg if steady[0] in k]:
            v = agg[k][-steady[1]:]
            agg['%s  [last %d launches: the iterations]' % (k, len(v))] = v
    tot = sum(sum(v) for v in agg.values())
    print('%-72s %6s %12s %12s %12s %12s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us',
                                                'max_us', '%'))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('%-72s %6d %12.1f %12.1f %12.1f %12.1f %6.2f' % (
            k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / tot))


def pmc(dbs, only='pass_kernel'):  # pca_xpass / pca_pass / gmm_pass
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute('select kernel_name, counter_name, value, duration, dispatch_id '
                           'from counters_collection').fetchall()
        per = defaultdict(lambda: defaultdict(float))
        dur = {}
        for n, c, v, d, disp in rows:
            if only and only not in n:
                continue
            per[(short(n), disp)][c] += v
            dur[(short(n), disp)] = d
        agg = defaultdict(lambda: defaultdict(list))
        for (k, disp), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
            agg[k]['duration_ns'].append(dur[(k, disp)])
        print('== %s' % db)
        for k, cs in agg.items():
            print(k)
            for c, v in sorted(cs.items()):
                print('    %-34s avg %18.1f   (n=%d, min %.1f, max %.1f)' % (
                    c, sum(v) / len(v), len(v), min(v), max(v)))


def timeline(db, count):
    """Last `count` kernel dispatches: start offset, duration, gap to the previous end (us)."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')").fetchall()]
    extra = [c for c in ('stream_id', 'queue_id') if c in cols]
    q = 'select name, start, end%s from kernels order by start' % ''.join(', ' + c for c in extra)
    rows = con.execute(q).fetchall()[-count:]
    t0 = rows[0][1]
    prev_end = None
    print('%-44s %12s %10s %10s  %s' % ('kernel', 'start_us', 'dur_us', 'gap_us', ' '.join(extra)))
    for r in rows:
        gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
        print('%-44s %12.1f %10.1f %10.1f  %s' % (short(r[0])[:44], (r[1] - t0) / 1e3,
                                                 (r[2] - r[1]) / 1e3, gap,
                                                 ' '.join(str(x) for x in r[3:])))
        prev_end = r[2] if prev_end is None else max(prev_end, r[2])


if __name__ == '__main__':
    args = sys.argv[1:]
    if args and args[0] == '--pmc':
        only = 'pass_kernel'
        rest = args[1:]
        if rest and rest[0] == '--only':            # substring of the kernel names to keep
            only, rest = rest[1], rest[2:]
        pmc(rest, only)
    elif args and args[0] == '--timeline':
        timeline(args[2], int(args[1]))
    else:
        steady = None
        if args and args[0] == '--steady':         # --steady <kernel substring> <n> db...
            steady, args = (args[1], int(args[2])), args[3:]
        for a in args:
            stats(a, steady)
