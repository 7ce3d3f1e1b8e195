"""BASELINE-sized units (C2 scaffolds: 5 Mbp x 5 Mbp, seed 50; C4 scaffolds: 10 Mbp x 10 Mbp, seed 1000) and the
whole C2 job (10 scaffolds, 100 ordered pairs) checked through size-independent properties,
because the CPU oracle needs minutes at this size: symmetry of the seed relation, validity of a
sample of hits against a numpy restatement of the seed rule, HSP scores recomputed on the host,
run-to-run determinism, and self-consistency of the full alignment stage."""
import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu

CARE = np.array([0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18])
HOX = np.array([[91, -114, -31, -123], [-114, 100, -125, -31], [-31, -125, 100, -114], [-123, -31, -114, 91]])


@pytest.fixture(scope='module', params=[(50, 5_000_000), (1000, 10_000_000)], ids=['c2_unit', 'c4_unit'])
def setup(request):
    from mimeo_amd import engine
    engine.init(0)
    seed, L = request.param
    names, seqs = synth_genome(seed, 2 * L, 2, repeat_frac=0.05)
    g = engine.Genome(names, seqs)
    code = [np.searchsorted(np.frombuffer(b'ACGT', np.uint8), s).astype(np.int8) for s in seqs]
    yield engine, g, seqs, code
    g.close()


def _key(h):
    return np.sort(h['tpos'].astype(np.uint64) << np.uint64(32) | h['qpos'].astype(np.uint64))


def test_seed_relation_is_symmetric_and_valid(setup):
    eng, g, seqs, code = setup
    ab = eng.seed_hits(g, 0, g, 1, 0)
    st = eng.stats()
    L = len(seqs[0])
    assert st['seed_hits'] == ab.size and ab.size > 13 * float(L) * L / 4 ** 12
    ba = eng.seed_hits(g, 1, g, 0, 0)
    swapped = ba['qpos'].astype(np.uint64) << np.uint64(32) | ba['tpos'].astype(np.uint64)
    assert np.array_equal(_key(ab), np.sort(swapped))
    # every sampled hit obeys the 12of19 + one-transition rule
    rng = np.random.default_rng(1)
    idx = rng.integers(0, ab.size, 20000)
    t = code[0][ab['tpos'][idx][:, None] + CARE[None, :]]
    q = code[1][ab['qpos'][idx][:, None] + CARE[None, :]]
    diff = t != q
    assert (diff.sum(1) <= 1).all()
    assert ((t ^ q)[diff] == 2).all()  # A<->G / C<->T flip only the high bit of the 2-bit code
    # and the expected number of random hits: 13 * Lt * Lq / 4^12, plus the planted repeats
    assert ab.size < 1.2 * 13 * float(L) * L / 4 ** 12


def test_hsps_deterministic_and_scores_recompute(setup):
    eng, g, seqs, code = setup
    p = eng.default_params(chain=0, entropy=0)
    h1 = eng.ungapped_hsps(g, 0, g, 1, 0, p)
    h2 = eng.ungapped_hsps(g, 0, g, 1, 0, p)
    assert h1.size > 100 and h1.size == h2.size
    cols = ['tstart', 'qstart', 'length', 'score', 'raw_score']
    assert np.array_equal(np.sort(h1[cols], order=cols), np.sort(h2[cols], order=cols))  # same set
    assert np.array_equal(h1, h2)  # and the documented order is total
    rng = np.random.default_rng(2)
    for k in rng.integers(0, h1.size, 300):
        h = h1[k]
        a = code[0][h['tstart']:h['tstart'] + h['length']]
        b = code[1][h['qstart']:h['qstart'] + h['length']]
        sc = HOX[a, b]
        assert int(sc.sum()) == int(h['raw_score']) >= 3000
        assert sc[0] > 0  # the left walk ends on a strict maximum, i.e. on a matching column


def test_minus_strand_equals_plus_strand_of_reverse_complement(setup):
    eng, g, seqs, code = setup
    comp = np.frombuffer(b'TGCA', np.uint8)[code[1]][::-1].copy()
    g2 = eng.Genome(['t', 'qrc'], [seqs[0], comp])
    a = eng.ungapped_hsps(g, 0, g, 1, 1, eng.default_params(chain=0))
    b = eng.ungapped_hsps(g2, 0, g2, 1, 0, eng.default_params(chain=0))
    assert a.size > 50 and np.array_equal(a, b)
    g2.close()


def test_full_alignment_stage_self_consistency(setup):
    eng, g, seqs, code = setup
    al = eng.align_pairs(g, None, [(0, 1), (1, 0), (0, 0)])
    st = eng.stats()
    assert st['pair_strands'] == 6 and st['alignments'] == al.size
    assert (al['score'] >= 3000).all() and (al['id_n'] <= al['id_d']).all()
    assert (al['tend'] - al['tstart'] >= al['id_d'] // 2).all()
    triv = al[(al['tid'] == 0) & (al['qid'] == 0) & (al['qstrand'] == 0)]
    L = len(seqs[0])
    assert triv.size == 1 and triv['tstart'][0] == 0 and triv['tend'][0] == L and triv['id_n'][0] == L
    # the planted repeats are found in both directions: (0,1) and (1,0) see mirror images
    ab = al[(al['tid'] == 0) & (al['qid'] == 1)]
    ba = al[(al['tid'] == 1) & (al['qid'] == 0)]
    assert ab.size > 10 and abs(int(ab.size) - int(ba.size)) <= max(3, ab.size // 10)


def test_self_scaffold_beyond_int32_score():
    """A 24 Mbp scaffold against itself: the trivial diagonal scores 2.3e9 > 2^31 (lastz's own score_t would
    wrap).  The gap-free stage saturates its 32-bit candidate score and k4_entropy recounts it in 64 bits; the
    identical-suffix shortcut of K6 carries a 64-bit score.  Checked against numpy: one full-length HSP and
    one full-length alignment with the exact HOXD70 sum."""
    from mimeo_amd import engine
    engine.init(0)
    rng = np.random.Generator(np.random.PCG64(24))
    L = 24_000_000
    code = rng.integers(0, 4, size=L, dtype=np.uint8)
    seq = np.frombuffer(b'ACGT', np.uint8)[code]
    exp = int(91 * L + 9 * int(np.count_nonzero((code == 1) | (code == 2))))
    assert exp > 2 ** 31
    g = engine.Genome(['big'], [seq])
    h = engine.ungapped_hsps(g, 0, g, 0, 0)
    full = h[(h['tstart'] == 0) & (h['qstart'] == 0) & (h['length'] == L)]
    assert full.size == 1 and int(full['raw_score'][0]) == exp and 0.99 * exp < int(full['score'][0]) <= exp
    a = engine.align_pair(g, 0, g, 0)
    triv = a[(a['tstart'] == 0) & (a['tend'] == L) & (a['qstart'] == 0) & (a['qend'] == L) & (a['qstrand'] == 0)]
    assert triv.size == 1 and int(triv['score'][0]) == exp and int(triv['id_n'][0]) == L == int(triv['id_d'][0])
    g.close()


def test_c2_whole_job_regions_sorted_disjoint_and_deterministic():
    """The whole C2 job (mimeo self, 50 Mbp, 10 scaffolds, --minIdt 80 --minLen 100 --minCov 3): 100 ordered
    pairs x 2 strands through K2-K6, the A11 filter and the coverage collapse.  Size-independent properties:
    every target's trivial self alignment is there, regions are sorted, disjoint, not book-ended, >= minLen and
    inside their scaffold, each is covered >= 3 deep by the kept records (recomputed on the host for a sample),
    and a second run returns the same bytes."""
    import bench
    from mimeo_amd import _ffi, engine
    engine.init(0)
    names, seqs = synth_genome(50, 50_000_000, 10)
    A = engine.Genome(names, seqs)
    pairs = [(t, q) for t in range(10) for q in range(10)]
    runs = []
    for _ in range(2):
        alns = engine.align_pairs(A, None, pairs)
        st = engine.stats()
        a = bench.a11_filter(alns, 100, 80)
        iv = np.zeros(a.size, dtype=_ffi.INTERVAL)
        iv['chrom'], iv['start'], iv['end'] = a['tid'], a['tstart'] + 1, a['tend']
        regions = engine.coverage_collapse(iv, [len(s) for s in seqs], 3, 100)
        runs.append((alns, regions, st))
    (alns, regions, st), (alns2, regions2, _) = runs
    assert alns.tobytes() == alns2.tobytes() and regions.tobytes() == regions2.tobytes()
    assert st['pair_strands'] == 200 and st['alignments'] == alns.size
    for t in range(10):
        triv = alns[(alns['tid'] == t) & (alns['qid'] == t) & (alns['qstrand'] == 0) & (alns['tstart'] == 0) & (alns['tend'] == 5_000_000)]
        assert triv.size == 1 and triv['id_n'][0] == 5_000_000
    assert regions.size > 100
    key = regions['chrom'].astype(np.int64) << 32 | regions['start']
    assert (np.diff(key) > 0).all()
    assert (regions['end'] - regions['start'] >= 100).all() and (regions['end'] <= 5_000_000).all()
    same = regions['chrom'][1:] == regions['chrom'][:-1]
    assert (regions['start'][1:][same] > regions['end'][:-1][same]).all()   # disjoint and not book-ended
    a = bench.a11_filter(alns, 100, 80)
    rng = np.random.default_rng(3)
    for k in rng.integers(0, regions.size, 40):
        r = regions[k]
        on = a[a['tid'] == r['chrom']]
        depth = np.zeros(int(r['end']) - int(r['start']) + 2, dtype=np.int32)   # positions start-1 .. end
        for s, e in zip(on['tstart'].astype(np.int64) + 1, on['tend'].astype(np.int64)):
            lo, hi = max(s, int(r['start']) - 1), min(e, int(r['end']) + 1)
            if lo < hi:
                depth[lo - (int(r['start']) - 1):hi - (int(r['start']) - 1)] += 1
        assert (depth[1:-1] >= 3).all() and depth[0] < 3 and depth[-1] < 3
    A.close()
