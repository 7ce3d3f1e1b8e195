#!/usr/bin/env python3
"""Randomised differential soak of the gap-free stage alone (K3 + K4 with its pre-filter) against the C oracle:
many marginal HSPs (diverged repeat copies, low thresholds, odd x-drops), N runs, soft masking, sequence ends.
Test infrastructure like tests/.   python scripts/soak_k4.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine  # noqa: E402
from mimeo_amd.synth import synth_genome  # noqa: E402
from oracle import oracle as O  # noqa: E402

HC = ['tstart', 'qstart', 'length', 'score', 'raw_score']


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    engine.init(0)
    t0, n, bad, nh = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        seed = int(rng.integers(1, 1 << 30))
        L = int(rng.choice([3_000, 30_000, 120_000, 400_000]))
        Lt = Lq = L
        # one case in sixteen is large enough for K34's tiles to hold several query segments (> 1280 entries: scaffolds beyond
        # 5.2 Mbp), or many target chunks per wavefront: the oracle needs seconds to a minute for these
        if rng.random() < 1.0 / 16:
            Lt, Lq = [(300_000, 8_000_000), (8_000_000, 300_000), (1_500_000, 12_000_000), (6_000_000, 6_000_000)][int(rng.choice([0, 0, 1, 1, 2, 3]))]
            L = max(Lt, Lq)
        big = L > 1_000_000   # (no microsatellites and fewer repeat copies there: the oracle walks every one of their seed hits)
        names, seqs = synth_genome(seed, 2 * L, 2, repeat_frac=float(rng.choice([0.0, 0.02, 0.05] if big else [0.0, 0.05, 0.3, 0.6])), families=int(rng.integers(1, 6)),
                                   cons_len=(40, int(rng.choice([120, 600, 3000]))), max_div=float(rng.choice([0.05, 0.25, 0.4])),
                                   indel_rate=float(rng.choice([0.0, 0.01, 0.05])), microsat_frac=0.0 if big else float(rng.choice([0.0, 0.0, 0.02])))
        seqs = [seqs[0][:Lt].copy(), seqs[1][:Lq].copy()]
        if rng.random() < 0.5:
            for s in seqs:
                for _ in range(int(rng.integers(0, 8))):
                    p = int(rng.integers(0, max(1, s.size - 300)))
                    s[p:p + int(rng.integers(1, 200))] = ord('N')
                for _ in range(int(rng.integers(0, 4))):
                    p = int(rng.integers(0, max(1, s.size - 2000)))
                    s[p:p + int(rng.integers(1, 1500))] |= 0x20
        kw = dict(transitions=int(rng.integers(0, 2)), entropy=int(rng.integers(0, 2)), chain=0,
                  hspthresh=int(rng.choice([800, 1200, 2000, 3000, 3000, 7000])), xdrop=int(rng.choice([500, 640, 910, 910, 1400, 4000])))
        tq = (0, 0) if (rng.random() < 0.25 and Lt == Lq) else (0, 1)
        g = engine.Genome(names, seqs)
        for strand in (0, 1):
            a = engine.ungapped_hsps(g, tq[0], g, tq[1], strand, engine.default_params(**kw))
            b = O.ungapped_hsps(seqs[tq[0]].tobytes(), seqs[tq[1]].tobytes(), strand, O.default_params(**kw))
            nh += b.size
            if not np.array_equal(np.sort(a[HC], order=HC), np.sort(b[HC], order=HC)):
                bad += 1
                print('HSP MISMATCH', seed, (Lt, Lq), kw, tq, strand, a.size, b.size, flush=True)
        g.close()
        n += 1
        if n % 50 == 0:
            print('cases', n, 'bad', bad, 'hsps', nh, ' %.0f s' % (time.time() - t0), flush=True)
    print('DONE cases', n, 'bad', bad, 'hsps compared', nh)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
