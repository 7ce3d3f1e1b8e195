#!/usr/bin/env python3
"""Randomised differential soak: HIP engine vs the C oracle on random small genomes and parameters
(development aid; test infrastructure like tests/).   python scripts/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine  # noqa: E402
from mimeo_amd.synth import synth_genome  # noqa: E402
from oracle import oracle as O  # noqa: E402

COLS = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
HC = ['tstart', 'qstart', 'length', 'score', 'raw_score']


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    engine.init(0)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        seed = int(rng.integers(1, 1 << 30))
        L = int(rng.choice([20_000, 60_000, 150_000, 300_000]))
        fam = int(rng.integers(1, 8))
        names, seqs = synth_genome(seed, 2 * L, 2, repeat_frac=float(rng.choice([0.02, 0.1, 0.3])), families=fam,
                                   cons_len=sorted((int(rng.choice([50, 300, 1500])), int(rng.choice([400, 2500, 8000])))) and (lambda a, b: (min(a, b), max(a, b) + 1))(int(rng.choice([50, 300, 1500])), int(rng.choice([400, 2500, 8000]))),
                                   max_div=float(rng.choice([0.0, 0.05, 0.15, 0.3])), indel_rate=float(rng.choice([0.0, 0.005, 0.03])),
                                   microsat_frac=float(rng.choice([0.0, 0.0, 0.01])))
        seqs = [s.copy() for s in seqs]
        for s in seqs:  # N runs and soft-masked stretches
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, s.size - 600))
                s[p:p + int(rng.integers(1, 500))] = ord('N')
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, s.size - 3000))
                s[p:p + int(rng.integers(1, 2500))] |= 0x20
        kw = dict(transitions=int(rng.integers(0, 2)), entropy=int(rng.integers(0, 2)), chain=int(rng.integers(0, 2)),
                  hspthresh=int(rng.choice([1500, 3000, 3000, 6000])), xdrop=int(rng.choice([500, 910, 910, 1500, 3000])),
                  ydrop=int(rng.choice([3400, 9400, 9400, 15000])), strand=int(rng.choice([1, 2, 3])),
                  gap_open=int(rng.choice([400, 400, 200, 1000])), gap_extend=int(rng.choice([30, 30, 10, 90])))
        self_pair = rng.random() < 0.3
        tq = (0, 0) if self_pair else (0, 1)
        g = engine.Genome(names, seqs)
        ok = True
        try:
            for strand in (0, 1):
                if not (kw['strand'] >> strand) & 1:
                    continue
                pk = dict(kw, chain=0)
                a = engine.ungapped_hsps(g, tq[0], g, tq[1], strand, engine.default_params(**pk))
                b = O.ungapped_hsps(seqs[tq[0]].tobytes(), seqs[tq[1]].tobytes(), strand, O.default_params(**pk))
                if not np.array_equal(np.sort(a[HC], order=HC), np.sort(b[HC], order=HC)):
                    ok = False
                    print('HSP MISMATCH', seed, L, kw, tq, strand, a.size, b.size, flush=True)
            tg = time.time()
            a = engine.align_pair(g, tq[0], g, tq[1], engine.default_params(**kw))
            tg = time.time() - tg
            tc = time.time()
            b = O.align_pair(seqs[tq[0]].tobytes(), seqs[tq[1]].tobytes(), O.default_params(**kw))
            tc = time.time() - tc
            if tg > 1.0 or tc > 20.0:
                print('slow case: gpu %.2f s  oracle %.1f s' % (tg, tc), seed, L, kw, tq, flush=True)
            if not np.array_equal(np.sort(a[COLS], order=COLS), np.sort(b[COLS], order=COLS)):
                ok = False
                print('ALIGN MISMATCH', seed, L, kw, tq, a.size, b.size, flush=True)
            if n % 5 == 0 and L <= 60_000:  # the multi-unit pipeline (lanes, batches) against per-pair oracle runs
                allp = [(0, 0), (0, 1), (1, 0), (1, 1)]
                aa = engine.align_pairs(g, None, allp, engine.default_params(**kw))
                for t, q in allp:
                    e = O.align_pair(seqs[t].tobytes(), seqs[q].tobytes(), O.default_params(**kw))
                    sub = aa[(aa['tid'] == t) & (aa['qid'] == q)]
                    if not np.array_equal(np.sort(sub[COLS], order=COLS), np.sort(e[COLS], order=COLS)):
                        ok = False
                        print('PAIRS MISMATCH', seed, L, kw, (t, q), sub.size, e.size, flush=True)
        except RuntimeError as e:
            if 'band' in str(e) or 'not supported' in str(e):
                print('limit', seed, L, kw, str(e)[:80], flush=True)
            else:
                ok = False
                print('ERROR', seed, L, kw, tq, e, flush=True)
        g.close()
        n += 1
        bad += 0 if ok else 1
        if n % 10 == 0:
            print('cases %d bad %d  %.0f s' % (n, bad, time.time() - t0), flush=True)
    print('DONE cases %d bad %d' % (n, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
