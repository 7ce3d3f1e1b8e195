#!/usr/bin/env python3
"""Randomised differential soak of the super-scaffold path (pack.hip): random fragmented genomes (uneven scaffold sizes down
to a few bases, N runs at the ends and inside, soft-masked stretches, planted repeats and microsatellites), random
scoring parameters, random packing geometry; the packed call against the unit-per-pair call (MIMEO_PACK=0), byte for
byte.   python scripts/soak_pack.py [seconds] [seed]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine  # noqa: E402
from mimeo_amd.synth import synth_genome, make_families  # noqa: E402


def fragmented(rng, seed, nscaf, mean_len, fams=None):
    names, seqs = synth_genome(seed, nscaf * mean_len, nscaf, repeat_frac=float(rng.choice([0.05, 0.2, 0.4])), families=int(rng.integers(1, 6)),
                               cons_len=(100, int(rng.choice([600, 2000]))), max_div=float(rng.choice([0.0, 0.1, 0.25])),
                               indel_rate=float(rng.choice([0.0, 0.01])), microsat_frac=float(rng.choice([0.0, 0.0, 0.02])), shared_families=fams)
    out = []
    for s in seqs:
        s = s[:max(1, int(s.size * rng.random() ** 0.5))].copy()
        if rng.random() < 0.15:
            s = s[:int(rng.integers(1, 40))].copy()
        for _ in range(int(rng.integers(0, 3))):
            if s.size > 50:
                p = int(rng.integers(0, s.size - 1))
                s[p:p + int(rng.integers(1, 120))] = ord('N')
        if rng.random() < 0.2:
            s[:int(rng.integers(1, 60))] = ord('N')
        if rng.random() < 0.2:
            s[-int(rng.integers(1, 60)):] = ord('N')
        if rng.random() < 0.3 and s.size > 200:
            p = int(rng.integers(0, s.size - 100))
            s[p:p + int(rng.integers(1, 3000))] |= 0x20
        out.append(s)
    return names, out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    engine.init(0)
    knobs = ('MIMEO_PACK', 'MIMEO_PACK_MIN', 'MIMEO_PACK_SUPER', 'MIMEO_PACK_MEMBER')
    t0, n, bad, packed_calls = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        seed = int(rng.integers(1, 1 << 30))
        nscaf = int(rng.integers(2, 40))
        mean_len = int(rng.choice([300, 2000, 8000, 30000]))
        kw = dict(transitions=int(rng.integers(0, 2)), entropy=int(rng.integers(0, 2)), chain=int(rng.integers(0, 2)),
                  hspthresh=int(rng.choice([1500, 3000, 3000, 6000])), xdrop=int(rng.choice([500, 910, 910, 3000])),
                  ydrop=int(rng.choice([3400, 9400, 9400])), strand=int(rng.choice([1, 2, 3, 3])))
        p = engine.default_params(**kw)
        mode = int(rng.integers(0, 3))
        if mode == 0:      # self, all ordered pairs
            names, seqs = fragmented(rng, seed, nscaf, mean_len)
            A, B = engine.Genome(names, seqs), None
            pairs = [(t, q) for t in range(nscaf) for q in range(nscaf)]
        elif mode == 1:    # a share of a self job: some targets against all
            names, seqs = fragmented(rng, seed, nscaf, mean_len)
            A, B = engine.Genome(names, seqs), None
            ts = sorted(set(int(x) for x in rng.integers(0, nscaf, size=max(1, nscaf // 3))))
            pairs = [(t, q) for t in ts for q in range(nscaf)]
        else:              # two genomes
            fams = make_families(np.random.Generator(np.random.PCG64(seed)), 4, (100, 1500))
            na, sa = fragmented(rng, seed, nscaf, mean_len, fams)
            nb_ = int(rng.integers(1, 12))
            nb, sb = fragmented(rng, seed + 1, nb_, mean_len, fams)
            A, B = engine.Genome(na, sa), engine.Genome(nb, sb)
            pairs = [(t, q) for t in range(nscaf) for q in range(nb_)]
        try:
            for k in knobs:
                os.environ.pop(k, None)
            os.environ['MIMEO_PACK'] = '0'
            ref = engine.align_pairs(A, B, pairs, p)
            os.environ.pop('MIMEO_PACK')
            os.environ['MIMEO_PACK_MIN'] = '2'
            os.environ['MIMEO_PACK_SUPER'] = str(int(rng.choice([5000, 40000, 300000, 8 << 20])))
            os.environ['MIMEO_PACK_MEMBER'] = str(int(rng.choice([1000, 10000, 2 << 20])))
            got = engine.align_pairs(A, B, pairs, p)
            st = engine.stats()
            packed_calls += st['super_units'] > 0
            if hashlib.md5(got.tobytes()).hexdigest() != hashlib.md5(ref.tobytes()).hexdigest():
                bad += 1
                print('MISMATCH', seed, nscaf, mean_len, mode, kw, {k: os.environ.get(k) for k in knobs}, got.size, ref.size, flush=True)
        except RuntimeError as e:
            bad += 1
            print('ERROR', seed, nscaf, mean_len, mode, kw, {k: os.environ.get(k) for k in knobs}, str(e)[:200], flush=True)
        finally:
            A.close()
            if B is not None:
                B.close()
        n += 1
        if n % 25 == 0:
            print('cases %d packed %d bad %d  %.0f s' % (n, packed_calls, bad, time.time() - t0), flush=True)
    print('DONE cases %d packed %d bad %d' % (n, packed_calls, bad), flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
