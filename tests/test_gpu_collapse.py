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


def test_bedgraph_known_answers(eng):
    """`bedtools genomecov -bg` on cases whose answer follows from its definition (per-base depth, runs of equal depth > 0;
    hand-derived like tests/golden/collapse_kat.json, not bedtools output: the image has no bedtools)."""
    cases = [
        ([(0, 0, 10), (0, 5, 15)], [100], [(0, 0, 5, 1), (0, 5, 10, 2), (0, 10, 15, 1)]),
        ([(0, 0, 10), (0, 10, 20)], [100], [(0, 0, 20, 1)]),                       # book-ended: one run of depth 1
        ([(0, 0, 10), (0, 0, 10), (0, 10, 20), (0, 10, 20)], [100], [(0, 0, 20, 2)]),
        ([(0, 5, 8), (1, 0, 3), (0, 20, 30)], [25, 10], [(0, 5, 8, 1), (0, 20, 25, 1), (1, 0, 3, 1)]),   # clipped to the chromosome
        ([(0, 7, 7), (0, 9, 3)], [100], []),                                       # empty and inverted intervals
    ]
    for iv, lens, want in cases:
        got = eng.coverage_bedgraph(np.asarray(iv, dtype=np.uint32), lens)
        assert [(int(r['chrom']), int(r['start']), int(r['end']), int(r['depth'])) for r in got] == want, (iv, got)


@pytest.mark.parametrize('seed,n', [(11, 3000), (12, 150000)])
def test_bedgraph_random_vs_per_base_oracle(eng, seed, n):
    from oracle import pipeline as P
    rng = np.random.default_rng(seed)
    lens = [40_000, 90_000, 555]
    chrom = rng.integers(0, 3, n)
    start = np.array([rng.integers(0, lens[c]) for c in chrom])
    ln = rng.integers(0, 2000, n)
    iv = np.stack([chrom, start, start + ln], 1).astype(np.uint32)
    got = eng.coverage_bedgraph(iv, lens)
    names = ['c0', 'c1', 'c2']
    exp = P.genomecov_bg([(names[c], s, e) for c, s, e in iv.tolist()], dict(zip(names, lens)))
    assert [(names[r['chrom']], int(r['start']), int(r['end']), int(r['depth'])) for r in got] == exp
    # the reference's chain of commands on top of it (awk $4 >= cov | sort | bedtools merge | awk len) is the fused collapse
    for cov in (1, 3):
        kept = [(c, s, e) for c, s, e, d in exp if d >= cov]
        merged = P.coverage_collapse(kept, dict(zip(names, lens)), 1, 100)
        fused = eng.coverage_collapse(iv, lens, cov, 100)
        assert [(names[r['chrom']], int(r['start']), int(r['end'])) for r in fused] == merged


def test_bedtools_shim_runs_the_reference_invocations(eng, tmp_path):
    """scripts/bedtools: the two command lines of src/mimeo/wrappers.py:1131-1150 through the device, output as bedtools prints it."""
    import subprocess, sys
    from oracle import pipeline as P
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(5)
    names, lens = ['chrB', 'chrA_2', 'scaf10'], [30_000, 8_000, 12_345]
    rows = []
    for _ in range(4000):
        c = int(rng.integers(0, 3))
        s = int(rng.integers(0, lens[c]))
        rows.append((names[c], s, min(lens[c], s + int(rng.integers(1, 900)))))
    rows.sort(key=lambda r: (r[0].encode(), r[1], r[2]))          # sort -k 1,1 -k 2n,3n
    bed, gen = tmp_path / 'temp_sorted.bed', tmp_path / 'lens.txt'
    bed.write_text(''.join('%s\t%d\t%d\n' % r for r in rows))
    gen.write_text(''.join('%s\t%d\n' % (n, l) for n, l in zip(names, lens)))
    r = subprocess.run([sys.executable, '-m', 'mimeo_amd.bedtools_shim', 'genomecov', '-bg', '-i', str(bed), '-g', str(gen)], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    exp = P.genomecov_bg(rows, dict(zip(names, lens)))
    assert r.stdout == ''.join('%s\t%d\t%d\t%d\n' % e for e in exp)
    cov3 = [e[:3] for e in exp if e[3] >= 3]
    bed2 = tmp_path / 'cov.bed'
    bed2.write_text(''.join('%s\t%d\t%d\n' % e for e in cov3))
    r = subprocess.run([sys.executable, '-m', 'mimeo_amd.bedtools_shim', 'merge', '-i', str(bed2)], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout == ''.join('%s\t%d\t%d\n' % e for e in P.coverage_collapse(rows, dict(zip(names, lens)), 3, 0))
    bad = subprocess.run([sys.executable, '-m', 'mimeo_amd.bedtools_shim', 'intersect', '-a', 'x', '-b', 'y'], cwd=root, capture_output=True, text=True)
    assert bad.returncode != 0 and 'unsupported' in bad.stderr
