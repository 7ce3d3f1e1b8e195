import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
engine.init(0)
names, seqs = synth_genome(50, 2 * L, 2, repeat_frac=0.05)
t = time.time(); g = engine.Genome(names, seqs); print('genome create s', time.time() - t)
for it in range(3):
    for strand in (0, 1):
        t = time.time(); h = engine.seed_hits(g, 0, g, 1, strand); dt = time.time() - t
        s = engine.stats()
        Lq = L; H = s['seed_hits']
        balg = (Lq + 3) // 4 + 8 * 13 * (Lq - 18) + 12 * H
        bker = 2 * 4 * (2**24 + 1) + 4 * 2 * L + 8 * H
        print('strand', strand, 'hits', H, 'wall %.3f s' % dt, 'ms_index %.2f ms_scan %.3f ms_fill %.3f' % (s['ms_index'], s['ms_scan'], s['ms_scan_fill']),
              'alg GB/s %.0f' % (balg / s['ms_scan'] / 1e6), 'fill-kernel own GB/s %.0f' % (bker / s['ms_scan_fill'] / 1e6))
