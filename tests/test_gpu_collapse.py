"""K7 parity: integer coverage collapse against the hand-derived known answers (SURVEY
Appendix B), the per-base Python oracle on random intervals, and size-independent properties."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def test_collapse_known_answers(eng):
    with open(os.path.join(G, 'collapse_kat.json')) as f:
        cases = json.load(f)['cases']
    for case in cases:
        names = sorted(case['chromlens'], key=lambda s: s.encode())
        cid = {n: i for i, n in enumerate(names)}
        iv = [(cid[c], s, e) for c, s, e in case['intervals']]
        got = eng.coverage_collapse(np.array(iv, dtype=np.uint32).reshape(-1, 3),
                                    [case['chromlens'][n] for n in names], case['min_cov'], case['min_len'])
        got = [[names[r['chrom']], int(r['start']), int(r['end'])] for r in got]
        assert got == case['expect'], case['name']


@pytest.mark.parametrize('seed,n,cov', [(1, 5000, 3), (2, 200000, 5), (3, 1000, 1)])
def test_collapse_random_vs_per_base_oracle(eng, seed, n, cov):
    from oracle import pipeline as P
    rng = np.random.default_rng(seed)
    lens = [50_000, 120_000, 777]
    chrom = rng.integers(0, 3, n)
    start = np.array([rng.integers(0, lens[c]) for c in chrom])
    ln = rng.integers(0, 3000, n)
    iv = np.stack([chrom, start, start + ln], 1).astype(np.uint32)
    got = eng.coverage_collapse(iv, lens, cov, 100)
    names = ['c0', 'c1', 'c2']
    exp = P.coverage_collapse([(names[c], s, e) for c, s, e in iv.tolist()], dict(zip(names, lens)), cov, 100)
    assert [(names[r['chrom']], int(r['start']), int(r['end'])) for r in got] == exp


def test_collapse_properties_large(eng):
    """Full-size properties: idempotence under min_cov=1 and sortedness / disjointness."""
    rng = np.random.default_rng(7)
    n = 3_000_000
    lens = [10_000_000] * 8
    chrom = rng.integers(0, 8, n).astype(np.uint32)
    start = rng.integers(0, 10_000_000 - 5000, n).astype(np.uint32)
    iv = np.stack([chrom, start, start + rng.integers(100, 5000, n).astype(np.uint32)], 1)
    r1 = eng.coverage_collapse(iv, lens, 3, 100)
    assert r1.size > 0
    k = r1['chrom'].astype(np.int64) << 32 | r1['start']
    assert np.all(np.diff(k) > 0)
    same = r1['chrom'][1:] == r1['chrom'][:-1]
    assert np.all(r1['start'][1:][same] > r1['end'][:-1][same])  # merged regions never touch
    r2 = eng.coverage_collapse(np.stack([r1['chrom'], r1['start'], r1['end']], 1), lens, 1, 100)
    assert np.array_equal(r1, r2)
