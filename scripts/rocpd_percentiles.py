#!/usr/bin/env python3
"""Per-kernel duration percentiles (us) of the library's kernels from a rocprofv3 rocpd database."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
d = collections.defaultdict(list)
for name, dur in db.execute('select name, duration from kernels'):
    m = re.search(r'mimeo::(\w+)', name)
    d[m.group(1) if m else ('rocprim' if 'rocprim' in name else name[:40])].append(dur / 1e3)
print('%-24s %5s %9s %8s %8s %8s %9s %9s' % ('kernel', 'n', 'sum_ms', 'min', 'p50', 'p90', 'p99', 'max'))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    n = len(v)
    print('%-24s %5d %9.1f %8.1f %8.1f %8.1f %9.1f %9.1f' % (k, n, sum(v) / 1e3, v[0], v[n // 2], v[int(n * .9)], v[min(n - 1, int(n * .99))], v[-1]))
