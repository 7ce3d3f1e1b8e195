"""K5/K6 parity (chain + gapped y-drop extension) and whole `lastz T Q`
parity through mimeo_align_pair / mimeo_align_pairs, against the C oracle."""
import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu

COLS = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _cmp(got, exp, tag, ordered=False):
    a, b = got[COLS], exp[COLS]
    if not ordered:
        a, b = np.sort(a, order=COLS), np.sort(b, order=COLS)
    assert a.size == b.size, (tag, a.size, b.size, a[:5], b[:5])
    bad = np.flatnonzero(a != b)
    assert bad.size == 0, (tag, a[bad[:5]], b[bad[:5]])


def test_chain_flags_match_oracle(eng):
    from oracle import oracle as O
    names, seqs = synth_genome(61, 400_000, 2, repeat_frac=0.2, families=3, cons_len=(300, 3000), max_div=0.1)
    g = eng.Genome(names, seqs)
    # the chained subset shows up as the set of alignments when gapped extension is off
    for strand, sbit in ((0, 1), (1, 2)):
        got = eng.align_pair(g, 0, g, 1, eng.default_params(gapped=0, strand=sbit))
        exp = O.ungapped_hsps(seqs[0].tobytes(), seqs[1].tobytes(), strand, O.default_params())
        exp = exp[(exp['flags'] & 1) == 1]
        assert got.size == exp.size and got.size > 0
        a = np.sort(np.stack([got['tstart'], got['tend'] - got['tstart'], got['score']], 1), axis=0)
        b = np.sort(np.stack([exp['tstart'], exp['length'], exp['score']], 1), axis=0)
        assert np.array_equal(a, b)
    g.close()


@pytest.mark.parametrize('seed,div,indel,chain', [(71, 0.15, 0.005, 1), (72, 0.05, 0.02, 1), (73, 0.12, 0.01, 0)])
def test_align_pair_matches_oracle(eng, seed, div, indel, chain):
    from oracle import oracle as O
    names, seqs = synth_genome(seed, 300_000, 2, repeat_frac=0.15, families=4, cons_len=(300, 2500),
                               max_div=div, indel_rate=indel)
    g = eng.Genome(names, seqs)
    got = eng.align_pair(g, 0, g, 1, eng.default_params(chain=chain))
    exp = O.align_pair(seqs[0].tobytes(), seqs[1].tobytes(), O.default_params(chain=chain))
    assert exp.size > 3
    _cmp(got, exp, (seed, chain), ordered=True)
    g.close()


def test_align_self_pair_trivial_alignment(eng):
    from oracle import oracle as O
    names, seqs = synth_genome(81, 60_000, 1, repeat_frac=0.1, families=2, cons_len=(300, 1500))
    g = eng.Genome(names, seqs)
    got = eng.align_pair(g, 0, g, 0)
    exp = O.align_pair(seqs[0].tobytes(), seqs[0].tobytes())
    _cmp(got, exp, 'self', ordered=True)
    assert got[0]['tend'] - got[0]['tstart'] == 60_000 and got[0]['id_n'] == 60_000
    g.close()


def test_align_pairs_self_mode_all_pairs(eng):
    """`mimeo self`: S^2 ordered pairs incl. (A,A); results concatenated in pair order."""
    from oracle import oracle as O
    names, seqs = synth_genome(91, 240_000, 3, repeat_frac=0.15, families=3, cons_len=(300, 2000))
    g = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(3) for q in range(3)]
    got = eng.align_pairs(g, None, pairs)
    st = eng.stats()
    assert st['pair_strands'] == 18 and st['seed_hits'] > 0 and st['alignments'] == got.size
    exp_all = []
    for t, q in pairs:
        e = O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        e['tid'], e['qid'] = t, q
        exp_all.append(e)
    exp = np.concatenate(exp_all)
    _cmp(got, exp, 'self-all', ordered=True)
    assert np.array_equal(got['tid'], exp['tid']) and np.array_equal(got['qid'], exp['qid'])
    g.close()


def test_align_with_n_and_lowercase(eng):
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    names, seqs = synth_genome(95, 200_000, 2, repeat_frac=0.2, families=3, cons_len=(500, 2500), max_div=0.08)
    T, Q = seqs[0].copy(), seqs[1].copy()
    for s in (T, Q):
        for _ in range(30):
            p = int(rng.integers(0, s.size - 400))
            s[p:p + int(rng.integers(1, 40))] = ord('N')
        for _ in range(30):
            p = int(rng.integers(0, s.size - 400))
            s[p:p + int(rng.integers(1, 200))] |= 0x20
    g = eng.Genome(['t', 'q'], [T, Q])
    got = eng.align_pair(g, 0, g, 1)
    exp = O.align_pair(T.tobytes(), Q.tobytes())
    _cmp(got, exp, 'masked', ordered=True)
    g.close()


def test_wide_bands_all_kernels(eng):
    """A large y-drop widens the DP band past the 1024-column register window (the 2048-column kernel takes over)
    and then past 2048 columns (the global-memory kernel takes over): identical results each time."""
    from oracle import oracle as O
    names, seqs = synth_genome(97, 120_000, 2, repeat_frac=0.2, families=2, cons_len=(800, 2000), max_div=0.1)
    g = eng.Genome(names, seqs)
    for yd in (25000, 90000):
        got = eng.align_pair(g, 0, g, 1, eng.default_params(ydrop=yd))
        exp = O.align_pair(seqs[0].tobytes(), seqs[1].tobytes(), O.default_params(ydrop=yd))
        assert exp.size > 2
        _cmp(got, exp, ('wide', yd), ordered=True)
    g.close()


def test_tandem_arrays_default_parameters(eng):
    """Tandem arrays make the live band as wide as the array (every shift by a period scores almost as well):
    with the DEFAULT parameters the register kernels overflow and the global-memory DP must produce exactly the
    oracle's alignments."""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)

    def array(unit, n, div):
        a = np.tile(unit, n)
        m = rng.random(a.size) < div
        a[m] = (a[m] + rng.integers(1, 4, int(m.sum()))) & 3
        return a
    u1, u2 = rng.integers(0, 4, 7), rng.integers(0, 4, 31)
    T = acgt[np.concatenate([rng.integers(0, 4, 4000), array(u1, 700, 0.03), rng.integers(0, 4, 3000), array(u2, 150, 0.05),
                             rng.integers(0, 4, 4000)])]
    Q = acgt[np.concatenate([rng.integers(0, 4, 2000), array(u2, 120, 0.05), rng.integers(0, 4, 5000), array(u1, 600, 0.03),
                             rng.integers(0, 4, 2500)])]
    g = eng.Genome(['t', 'q'], [T, Q])
    got = eng.align_pair(g, 0, g, 1)
    exp = O.align_pair(T.tobytes(), Q.tobytes())
    assert exp.size >= 2
    _cmp(got, exp, 'tandem', ordered=True)
    g.close()


@pytest.mark.parametrize('cap', [None, '100000'])
def test_long_extension_beyond_packed_counts(eng, monkeypatch, cap):
    """A 300 kb alignment with 3 % substitutions and a few indels: each half extension runs for more than
    65535 rows, past what the packed match/mismatch counters of the four-wavefront DP kernel hold, so the
    single-wavefront kernel must take over — same alignment, same identity counts as the oracle.

    cap = 100000: the score beyond which the 2048-column kernel hands a half extension to k6_dp_any, and beyond which that
    kernel moves its 32-bit cells down (2 * 10^9 in production: 20 Mbp of near-identity in one alignment; the oracle
    scores in 64 bits).  Here every half extension of the long alignment is rebased some 250 times: same alignment."""
    from oracle import oracle as O
    if cap:
        monkeypatch.setenv('MIMEO_K6_SCORE_CAP', cap)
    rng = np.random.default_rng(123)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)
    core = rng.integers(0, 4, 300_000)
    mut = core.copy()
    sub = rng.random(core.size) < 0.03
    mut[sub] = (mut[sub] + rng.integers(1, 4, int(sub.sum()))) & 3
    for p in sorted(rng.integers(1000, core.size - 1000, 40).tolist(), reverse=True):  # indels of 1-3 bases
        mut = np.delete(mut, slice(p, p + int(rng.integers(1, 4)))) if rng.random() < 0.5 else np.insert(mut, p, rng.integers(0, 4, int(rng.integers(1, 4))))
    T = acgt[np.concatenate([rng.integers(0, 4, 5000), core, rng.integers(0, 4, 5000)])]
    Q = acgt[np.concatenate([rng.integers(0, 4, 3000), mut, rng.integers(0, 4, 3000)])]
    g = eng.Genome(['t', 'q'], [T, Q])
    p = eng.default_params(strand=1)
    got = eng.align_pair(g, 0, g, 1, p)
    exp = O.align_pair(T.tobytes(), Q.tobytes(), O.default_params(strand=1))
    assert exp.size >= 1 and int((exp['tend'] - exp['tstart']).max()) > 280_000
    _cmp(got, exp, 'long', ordered=True)
    g.close()


def test_chain_with_tens_of_thousands_of_hsps(eng, monkeypatch):
    """K5 at a size where its structure matters: ~7e4 HSPs in ONE unit (a low hspthresh on an 800 kbp pair makes
    nearly every strong seed hit an HSP): the wave kernel for large groups (k5_chain_wave: a Fenwick tree over query ends).
    The chained subset — the alignments with gapped extension off — must equal the oracle's O(n^2) chain, and the
    anchor order (one stable device-wide sort by group, chained, score) must be the oracle's: score descending, then
    (tstart, qstart, length).  The one-level kernel (MIMEO_K5_NO_BIG) and the two-level kernel (MIMEO_K5_BIG=old: blocks of
    2048 HSPs, the path of groups beyond 2^24 HSPs) must give the same."""
    from oracle import oracle as O
    names, seqs = synth_genome(66, 1_600_000, 2, repeat_frac=0.1, families=3, cons_len=(300, 3000), max_div=0.1)
    g = eng.Genome(names, seqs)
    got = eng.align_pair(g, 0, g, 1, eng.default_params(gapped=0, strand=1, hspthresh=1150, entropy=0))
    st = eng.stats()
    assert st['hsps'] > 50_000
    exp = O.ungapped_hsps(seqs[0].tobytes(), seqs[1].tobytes(), 0, O.default_params(hspthresh=1150, entropy=0))
    assert exp.size == st['hsps']
    exp = exp[(exp['flags'] & 1) == 1]
    assert got.size == exp.size > 300
    order = np.lexsort((exp['length'], exp['qstart'], exp['tstart'], -exp['score']))   # anchor order
    e = exp[order]
    assert np.array_equal(got['tstart'], e['tstart']) and np.array_equal(got['qstart'], e['qstart'])
    assert np.array_equal(got['tend'] - got['tstart'], e['length']) and np.array_equal(got['score'], e['score'])
    monkeypatch.setenv('MIMEO_K5_NO_BIG', '1')
    one_level = eng.align_pair(g, 0, g, 1, eng.default_params(gapped=0, strand=1, hspthresh=1150, entropy=0))
    monkeypatch.delenv('MIMEO_K5_NO_BIG')
    assert one_level.tobytes() == got.tobytes()
    monkeypatch.setenv('MIMEO_K5_BIG', 'old')
    two_level = eng.align_pair(g, 0, g, 1, eng.default_params(gapped=0, strand=1, hspthresh=1150, entropy=0))
    monkeypatch.delenv('MIMEO_K5_BIG')
    assert two_level.tobytes() == got.tobytes()
    g.close()


def test_wave_chain_on_every_group_size(eng, monkeypatch):
    """MIMEO_K5_BIG_MIN=1 sends every group with two or more HSPs through the wave kernel: groups of a few HSPs (tiles only),
    collinear runs (every HSP a step's predecessor), tandem arrays and microsatellite rectangles (waves of many HSPs that
    start together) — chained subsets and alignments against the oracle, both strands."""
    from oracle import oracle as O
    monkeypatch.setenv('MIMEO_K5_BIG_MIN', '1')
    names, seqs = synth_genome(93, 900_000, 3, repeat_frac=0.3, families=3, cons_len=(200, 2500), max_div=0.2, indel_rate=0.02, microsat_frac=0.02)
    g = eng.Genome(names, seqs)
    nchained = 0
    for t, q in ((0, 1), (1, 2), (2, 2)):
        for strand, sbit in ((0, 1), (1, 2)):
            got = eng.align_pair(g, t, g, q, eng.default_params(gapped=0, strand=sbit, hspthresh=2000))
            exp = O.ungapped_hsps(seqs[t].tobytes(), seqs[q].tobytes(), strand, O.default_params(hspthresh=2000))
            exp = exp[(exp['flags'] & 1) == 1]
            e = exp[np.lexsort((exp['length'], exp['qstart'], exp['tstart'], -exp['score']))]
            assert got.size == e.size, (t, q, strand, got.size, e.size)
            assert np.array_equal(got['tstart'], e['tstart']) and np.array_equal(got['score'], e['score']) and np.array_equal(got['tend'] - got['tstart'], e['length'])
            nchained += e.size
    assert nchained > 100
    pairs = [(t, q) for t in range(3) for q in range(3)]
    got = eng.align_pairs(g, None, pairs)
    exp_all = []
    for t, q in pairs:
        e = O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        e['tid'], e['qid'] = t, q
        exp_all.append(e)
    _cmp(got, np.concatenate(exp_all), 'wave-all', ordered=True)
    g.close()
