#!/usr/bin/env python3
"""Soak (CPU): the C oracle against specification v1 in plain Python (tests/spec_v1.py) on random small pairs with random parameters.

    python scripts/soak_spec.py [seconds] [first seed]     -> one line per case, a summary line at the end

Per case: a target and a query of 300-1600 bases sharing mutated copies (substitutions, indels, some reverse-complemented) of a few
consensus sequences, N runs and soft-masked stretches at random, microsatellites now and then; random thresholds, x-drops, y-drops
and gap penalties.  Compared: seed hits and HSPs of both strands, the alignments of the pair.  PARITY UNPINNED (neither side is LASTZ):
this shows that two restatements of the documented rules agree."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O                     # noqa: E402
from tests import spec_v1 as S                     # noqa: E402
from tests.test_oracle_rules import _pair          # noqa: E402


def one(seed):
    rng = np.random.default_rng(seed)
    lt, lq = int(rng.integers(300, 1600)), int(rng.integers(300, 1600))
    kw = dict(sub=float(rng.choice([0.02, 0.06, 0.1, 0.15])), indel=float(rng.choice([0.0, 0.01, 0.03])), lower=bool(rng.random() < 0.4),
              ns=bool(rng.random() < 0.4), rc_copy=bool(rng.random() < 0.5))
    T, Q = _pair(seed, lt, lq, int(rng.integers(1, 6)), cons=(100, min(lt, lq) // 2), **kw)
    if rng.random() < 0.3:                                                     # a microsatellite on both
        unit = bytes(rng.choice(list(b'ACGT'), int(rng.integers(1, 5))).tolist())
        p, q = int(rng.integers(0, lt - 10)), int(rng.integers(0, lq - 10))
        T = T[:p] + unit * int(rng.integers(20, 70)) + T[p:]
        Q = Q[:q] + unit * int(rng.integers(20, 70)) + Q[q:]
    par = dict(hspthresh=int(rng.choice([3000, 3000, 2200, 5000])), xdrop=int(rng.choice([910, 910, 340, 1500])),
               ydrop=int(rng.choice([9400, 9400, 3000, 1500])), gap_open=int(rng.choice([400, 400, 200])),
               gap_extend=int(rng.choice([30, 30, 60])), transitions=int(rng.random() < 0.8), entropy=int(rng.random() < 0.8))
    bad = []
    for minus in (0, 1):
        q = S.revcomp(Q.decode()) if minus else Q.decode()
        got = O.seed_hits(T, Q, minus, O.default_params(**par))
        if [(int(a), int(b)) for a, b in zip(got['tpos'], got['qpos'])] != S.seed_hits(T.decode(), q, bool(par['transitions'])):
            bad.append('hits%d' % minus)
        got = O.ungapped_hsps(T, Q, minus, O.default_params(chain=0, **par))
        exp = S.ungapped_hsps(T.decode(), q, par['hspthresh'], par['xdrop'], bool(par['transitions']), bool(par['entropy']))
        if sorted((int(h['tstart']), int(h['qstart']), int(h['length']), int(h['score']), int(h['raw_score'])) for h in got) != sorted(exp):
            bad.append('hsps%d' % minus)
    n = 0
    if par['transitions'] and par['entropy']:                                  # spec_v1.align_strand takes the default seed / entropy rules
        got = O.align_pair(T, Q, O.default_params(**par))
        exp = []
        for minus in (0, 1):
            exp += S.align_strand(T.decode(), Q.decode(), minus, par['hspthresh'], par['xdrop'], par['ydrop'], par['gap_open'], par['gap_extend'])
        g = sorted((int(a['tstart']), int(a['tend']), int(a['qstart']), int(a['qend']), int(a['score']), int(a['id_n']), int(a['id_d']),
                    int(a['qstrand'])) for a in got)
        if g != sorted(exp):
            bad.append('alignments')
        n = len(exp)
    return bad, n, lt, lq


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, cases, fails, alns = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        bad, n, lt, lq = one(seed)
        cases += 1
        alns += n
        if bad:
            fails += 1
            print('MISMATCH seed %d (%d x %d): %s' % (seed, lt, lq, ' '.join(bad)), flush=True)
        seed += 1
    print('soak_spec: %d cases (seeds %d..%d), %d alignments compared, %d mismatching cases, %.0f s'
          % (cases, seed - cases, seed - 1, alns, fails, time.time() - t0), flush=True)
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
