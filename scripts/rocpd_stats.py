#!/usr/bin/env python3
"""Kernel summary (calls, total, average, share) from a rocprofv3 rocpd database (`--kernel-trace --stats`
writes `<name>_results.db` on ROCm 7) as a small CSV with shortened kernel names.
    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/<name>.csv"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'(mimeo::\w+(<\d+>)?)', name)
    if m:
        return m.group(1)
    m = re.search(r'detail::(\w+)<', name.split('trampoline_kernel<')[-1]) if 'rocprim' in name else None
    if m:
        kt = re.search(r'wrapped_\w+_config<[^,]+, ([\w ]+)', name)
        return 'rocprim::' + m.group(1).replace('wrapped_', '').replace('_config', '') + ('<%s>' % kt.group(1).strip() if kt else '')
    return name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    print('kernel,calls,total_ms,avg_us,percent')
    for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%s,%d,%.2f,%.2f,%.2f' % (k, calls, tot / 1e3, tot / calls, pct))  # the view reports microseconds


if __name__ == '__main__':
    main()
