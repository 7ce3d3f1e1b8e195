"""K4 parity: HSPs from the stateless-extension + segment-resolution kernels must equal the
oracle's sequential lastz-shaped scan, bit for bit."""
import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu

COLS = ['tstart', 'qstart', 'length', 'score', 'raw_score']


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _cmp(got, exp, tag):
    a = np.sort(got[COLS], order=COLS)
    b = np.sort(exp[COLS], order=COLS)
    assert a.size == b.size, (tag, a.size, b.size)
    bad = np.flatnonzero(a != b)
    assert bad.size == 0, (tag, a[bad[:5]], b[bad[:5]])


@pytest.mark.parametrize('seed,div,indel', [(21, 0.15, 0.005), (22, 0.02, 0.0), (23, 0.25, 0.02)])
def test_hsps_match_oracle_cross(eng, seed, div, indel):
    from oracle import oracle as O
    names, seqs = synth_genome(seed, 400_000, 2, repeat_frac=0.15, families=5, cons_len=(200, 3000),
                               max_div=div, indel_rate=indel)
    g = eng.Genome(names, seqs)
    for strand in (0, 1):
        got = eng.ungapped_hsps(g, 0, g, 1, strand, eng.default_params(chain=0))
        exp = O.ungapped_hsps(seqs[0].tobytes(), seqs[1].tobytes(), strand, O.default_params(chain=0))
        assert exp.size > 10
        _cmp(got, exp, (seed, strand))
    g.close()


def test_hsps_self_pair_trivial_diagonal(eng):
    """(A,A) without --self: the full-length diagonal is one HSP found by the long-walk kernel,
    and every other hit on it is a follower that must be skipped."""
    from oracle import oracle as O
    names, seqs = synth_genome(31, 300_000, 1, repeat_frac=0.1, families=3, cons_len=(300, 2000))
    g = eng.Genome(names, seqs)
    for strand in (0, 1):
        got = eng.ungapped_hsps(g, 0, g, 0, strand, eng.default_params(chain=0))
        exp = O.ungapped_hsps(seqs[0].tobytes(), seqs[0].tobytes(), strand, O.default_params(chain=0))
        _cmp(got, exp, ('self', strand))
        if strand == 0:
            assert got['length'].max() == 300_000
    g.close()


def test_hsps_self_pair_with_n_runs_and_soft_mask(eng):
    """The main diagonal of a self unit is replayed on the seed-validity planes (k4_diag0): N runs longer
    than the x-drop allows cut it into several HSPs, shorter ones are bridged, and soft-masked stretches have
    no target seeds — all of it must match the sequential oracle, and so must the full alignment."""
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    names, seqs = synth_genome(43, 400_000, 1, repeat_frac=0.1, families=3, cons_len=(300, 2000))
    S = seqs[0].copy()
    for ln in (1, 3, 8, 9, 10, 12, 40, 200, 5000):  # 9 Ns cost 900 < xdrop, 10 cost 1000 > xdrop
        p = int(rng.integers(1000, S.size - 6000))
        S[p:p + ln] = ord('N')
    for _ in range(20):
        p = int(rng.integers(0, S.size - 3000))
        S[p:p + int(rng.integers(20, 2500))] |= 0x20
    S[:50] = ord('N')
    S[-30:] |= 0x20
    g = eng.Genome(['s'], [S])
    for strand in (0, 1):
        got = eng.ungapped_hsps(g, 0, g, 0, strand, eng.default_params(chain=0))
        exp = O.ungapped_hsps(S.tobytes(), S.tobytes(), strand, O.default_params(chain=0))
        _cmp(got, exp, ('self-masked', strand))
    diag = got if False else eng.ungapped_hsps(g, 0, g, 0, 0, eng.default_params(chain=0))
    on_diag = diag[diag['tstart'] == diag['qstart']]
    assert on_diag.size >= 4  # the long N runs split the trivial diagonal
    a, b = eng.align_pair(g, 0, g, 0), O.align_pair(S.tobytes(), S.tobytes())
    cols = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
    assert np.array_equal(np.sort(a[cols], order=cols), np.sort(b[cols], order=cols))
    g.close()


def test_hsps_with_n_runs_lowercase_and_no_entropy(eng):
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    names, seqs = synth_genome(41, 300_000, 2, repeat_frac=0.2, families=3, cons_len=(500, 3000), max_div=0.05)
    T, Q = seqs[0].copy(), seqs[1].copy()
    for s in (T, Q):
        for _ in range(40):
            p = int(rng.integers(0, s.size - 400))
            s[p:p + int(rng.integers(1, 60))] = ord('N')
        for _ in range(40):
            p = int(rng.integers(0, s.size - 400))
            s[p:p + int(rng.integers(1, 300))] |= 0x20
    g = eng.Genome(['t', 'q'], [T, Q])
    for ent in (1, 0):
        for strand in (0, 1):
            got = eng.ungapped_hsps(g, 0, g, 1, strand, eng.default_params(chain=0, entropy=ent))
            exp = O.ungapped_hsps(T.tobytes(), Q.tobytes(), strand, O.default_params(chain=0, entropy=ent))
            _cmp(got, exp, ('mask', ent, strand))
    g.close()


def test_hsps_low_complexity(eng):
    """Microsatellites: many hits per diagonal, staggered diagonals, entropy adjustment bites."""
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    names, seqs = synth_genome(51, 120_000, 2, repeat_frac=0.0)
    T, Q = seqs[0].copy(), seqs[1].copy()
    for unit in (b'A', b'AC', b'AAG', b'ACGT', b'AAAAG', b'ACACGT'):
        for s in (T, Q):
            p = int(rng.integers(0, s.size - 2000))
            ln = int(rng.integers(200, 900))
            rep = np.frombuffer((unit * (ln // len(unit) + 1))[:ln], dtype=np.uint8)
            s[p:p + ln] = rep
    g = eng.Genome(['t', 'q'], [T, Q])
    for strand in (0, 1):
        got = eng.ungapped_hsps(g, 0, g, 1, strand, eng.default_params(chain=0))
        exp = O.ungapped_hsps(T.tobytes(), Q.tobytes(), strand, O.default_params(chain=0))
        _cmp(got, exp, ('ssr', strand))
    g.close()


def test_parameter_domain_is_enforced(eng):
    """xdrop below 4 x 125 would break the four-columns-per-step walk: the ABI refuses it instead of computing
    something else (found by scripts/soak.py); other nonsense is refused too."""
    names, seqs = synth_genome(3, 40_000, 2)
    g = eng.Genome(names, seqs)
    for kw, msg in ((dict(xdrop=340), 'xdrop'), (dict(gap_extend=0), 'out of range'), (dict(strand=0), 'strand')):
        with pytest.raises(RuntimeError, match=msg):
            eng.ungapped_hsps(g, 0, g, 1, 0, eng.default_params(**kw))
        with pytest.raises(RuntimeError, match=msg):
            eng.align_pairs(g, None, [(0, 1)], eng.default_params(**kw))
    assert eng.ungapped_hsps(g, 0, g, 1, 0, eng.default_params(xdrop=500)).size >= 0
    g.close()


def test_heavy_kernel_variants_give_identical_hsps(eng, monkeypatch):
    """Two independent decompositions of the stage must give byte-identical HSP sets: the fused seed-scan /
    pre-filter / walk kernel on seed frames (K34, production) and the round-1 pair "materialised hit array +
    K4 fast kernel" (MIMEO_HEAVY=v1) with its pre-filter on the two-plane copy (9), on the full planes (5) and
    without any pre-filter (1: every hit walked exactly).  A pre-filter may only drop hits whose exact walk
    yields nothing.  On N-free sequence and with N runs, cross and self units, both strands; and with the
    queues so small that the batch is repeated (MIMEO_QUEUE_SHRINK)."""
    names, seqs = synth_genome(91, 1_200_000, 2, repeat_frac=0.15, families=5, cons_len=(200, 3000), max_div=0.2)
    for mode in ('clean', 'n'):
        ss = [s.copy() for s in seqs]
        if mode == 'n':
            ss[0][5000:5300] = ord('N'); ss[1][70000:70010] = ord('N'); ss[1][300000:300001] = ord('N')
        g = eng.Genome(names, ss)
        res = {}
        for tag, env in (('fused', {}), ('v1', {'MIMEO_HEAVY': 'v1'}), ('v1_walk_all', {'MIMEO_HEAVY': 'v1', 'MIMEO_K4_VARIANT': '1'}),
                         ('v1_full_planes', {'MIMEO_HEAVY': 'v1', 'MIMEO_K4_VARIANT': '5'}), ('fused_rerun', {'MIMEO_QUEUE_SHRINK': '4000'})):
            for k in ('MIMEO_HEAVY', 'MIMEO_K4_VARIANT', 'MIMEO_QUEUE_SHRINK'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out, hits, reruns = [], 0, 0
            for t, q in ((0, 1), (0, 0)):
                for strand in (0, 1):
                    out.append(eng.ungapped_hsps(g, t, g, q, strand))
                    st = eng.stats()
                    hits += st['seed_hits']
                    reruns += st['queue_reruns']
            res[tag] = (b''.join(o.tobytes() for o in out), hits)
            assert (reruns > 0) == (tag == 'fused_rerun'), (tag, reruns)
        for k in ('MIMEO_HEAVY', 'MIMEO_K4_VARIANT', 'MIMEO_QUEUE_SHRINK'):
            monkeypatch.delenv(k, raising=False)
        assert len(res['fused'][0]) // 32 > 100
        for tag in res:
            assert res[tag] == res['fused'], (mode, tag, len(res[tag][0]) // 32, res[tag][1], res['fused'][1])
        g.close()


def test_tiny_scaffolds_and_hits_at_the_ends(eng):
    """Scaffolds of 19..300 bases carrying copies of one another flush with their ends: every frame of the
    pre-filter and of the exact walk hangs over a sequence end (zero padding), on both strands."""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    names, seqs = [], []
    core = rng.integers(0, 4, size=400, dtype=np.uint8)
    for k in range(24):
        n = int(rng.choice([19, 20, 37, 64, 83, 100, 147, 200, 300]))
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        m = int(rng.integers(19, n + 1))
        off = int(rng.integers(0, 400 - m + 1))
        piece = core[off:off + m].copy()
        if rng.random() < 0.5:
            piece = (3 - piece)[::-1]
        where = int(rng.choice([0, n - m]))          # flush with the start or with the end
        s[where:where + m] = piece
        for _ in range(int(rng.integers(0, 3))):     # a few substitutions inside
            s[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        names.append('t%d' % k)
        seqs.append(np.frombuffer(b'ACGT', np.uint8)[s])
    g = eng.Genome(names, seqs)
    cols = ['tstart', 'qstart', 'length', 'score', 'raw_score']
    nh = 0
    for kw in ({'hspthresh': 1000}, {'hspthresh': 1800, 'xdrop': 500}):
        for t in range(0, 24, 6):
            for q in range(24):
                for strand in (0, 1):
                    a = eng.ungapped_hsps(g, t, g, q, strand, eng.default_params(**kw))
                    b = O.ungapped_hsps(seqs[t].tobytes(), seqs[q].tobytes(), strand, O.default_params(**kw))
                    assert np.array_equal(np.sort(a[cols], order=cols), np.sort(b[cols], order=cols)), (kw, t, q, strand)
                    nh += b.size
    assert nh > 40
    g.close()
