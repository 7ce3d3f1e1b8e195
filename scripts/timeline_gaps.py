#!/usr/bin/env python3
"""Gaps between the heavy kernels of consecutive units in a rocprofv3 kernel trace (csv) of bench.py:
   python scripts/timeline_gaps.py gpurun_out/tl/tl_kernel_trace.csv"""
import collections
import csv
import re
import statistics
import sys


def short(n):
    m = re.search(r'mimeo::(\w+)', n)
    return m.group(1) if m else ('rocprim' if 'rocprim' in n else 'other')


rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id']) for r in rows)
k5 = [e for e in ev if e[2] == 'k5_chain']
t0, t1 = k5[-2][1], k5[-1][0]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
first = [e for e in win if e[2] == 'k3_join_count'][0][0]
win = [e for e in win if e[0] >= first]
span = (t1 - first) / 1e6
cur_s = cur_e = None
busy = 0
for s, e, _, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('K3/K4 phase of the last step: %.1f ms, %d kernels, GPU busy (union) %.1f%%' % (span, len(win), 100 * busy / 1e6 / span))
tot = collections.Counter()
for s, e, k, q in win:
    tot[k] += (e - s) / 1e6
print('  ' + ', '.join('%s %.1f' % kv for kv in tot.most_common(10)))
fast = sorted((s, e, q) for s, e, k, q in win if k == 'k4_extend_hits')
cnt = sorted((s, e, q) for s, e, k, q in win if k == 'k3_join_count')
fill = sorted((s, e, q) for s, e, k, q in win if k == 'k3_join_fill')
n = min(len(fast), len(cnt) - 1, len(fill))
g1 = [(cnt[i + 1][0] - fast[i][1]) / 1e3 for i in range(n)]
g2 = [(fill[i][0] - cnt[i][1]) / 1e3 for i in range(n)]
g3 = [(fast[i][0] - fill[i][1]) / 1e3 for i in range(n)]
for name, g in (('fast end -> next count start', g1), ('count end -> fill start', g2), ('fill end -> fast start', g3)):
    print('%-30s median %.1f us  mean %.1f us' % (name, statistics.median(g), statistics.mean(g)))
print('durations (median us): count %.1f fill %.1f fast %.1f' % tuple(statistics.median([(e - s) / 1e3 for s, e, _ in x]) for x in (cnt, fill, fast)))
# what runs in the gap after the fast kernel of unit i, and on which queue relative to the next count
inside = collections.Counter()
for i in range(n):
    a, b = fast[i][1], cnt[i + 1][0]
    for s, e, k, q in win:
        if e > a and s < b and k not in ('k4_extend_hits', 'k3_join_count'):
            inside[(k, 'same queue as next count' if q == cnt[i + 1][2] else 'other queue')] += (min(e, b) - max(s, a)) / 1e3
print('kernel time inside those gaps (us per unit):')
for k, v in inside.most_common(8):
    print('   %-22s %-26s %.1f' % (k[0], k[1], v / n))
