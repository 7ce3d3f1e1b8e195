"""K1-K3 parity: the HIP index join must produce exactly the oracle's seed-hit multiset
(SURVEY §8a A6/A7).  Calls go through the C-ABI (mimeo_amd._ffi)."""
import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _sorted_hits(h):
    k = h['tpos'].astype(np.uint64) << np.uint64(32) | h['qpos'].astype(np.uint64)
    return np.sort(k)


def _mask_some(seq, rng, nrun=6, lower=False):
    s = seq.copy()
    for _ in range(nrun):
        p = int(rng.integers(0, s.size - 200))
        ln = int(rng.integers(1, 150))
        if lower:
            s[p:p + ln] |= 0x20
        else:
            s[p:p + ln] = ord('N')
    return s


@pytest.mark.parametrize('case', ['plain', 'n_runs', 'lowercase', 'short', 'transitions_off'])
def test_seed_hits_match_oracle(eng, case):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    names, seqs = synth_genome(101, 300_000, 2, repeat_frac=0.1, families=4, cons_len=(300, 1500))
    T, Q = seqs[0], seqs[1]
    kw = {}
    if case == 'n_runs':
        T, Q = _mask_some(T, rng), _mask_some(Q, rng)
    elif case == 'lowercase':
        T, Q = _mask_some(T, rng, lower=True), _mask_some(Q, rng, lower=True)
    elif case == 'short':
        T, Q = T[:19], Q[:57]
        Q[20:39] = T  # one exact seed hit at (0, 20)
    elif case == 'transitions_off':
        kw = {'transitions': 0}
    g = eng.Genome(['t', 'q'], [T, Q])
    for strand in (0, 1):
        got = eng.seed_hits(g, 0, g, 1, strand, eng.default_params(**kw))
        exp = O.seed_hits(T.tobytes(), Q.tobytes(), strand, O.default_params(**kw))
        assert got.size == exp.size, (case, strand, got.size, exp.size)
        assert np.array_equal(_sorted_hits(got), _sorted_hits(exp)), (case, strand)
    g.close()


def test_empty_and_tiny_scaffolds(eng):
    g = eng.Genome(['a', 'b', 'c'], [b'', b'ACGT', b'ACGTACGTACGTACGTACGTACGT'])
    for t in range(3):
        for q in range(3):
            for strand in (0, 1):
                h = eng.seed_hits(g, t, g, q, strand)
                if t == 2 and q == 2 and strand == 0:
                    assert h.size > 0
                if t < 2 or q < 2:
                    assert h.size == 0
    g.close()


def test_self_diagonal_present(eng):
    names, seqs = synth_genome(5, 100_000, 1, repeat_frac=0.0)
    g = eng.Genome(names, seqs)
    h = eng.seed_hits(g, 0, g, 0, 0)
    diag = h[h['tpos'] == h['qpos']]
    assert diag.size == 100_000 - 18
    g.close()


@pytest.mark.parametrize('big_side', ['target', 'query'])
def test_tile_with_more_than_65535_entries(eng, big_side):
    """A satellite array (here 70 kb of poly-A: 70 000 entries under ONE key) on either side: far more
    entries in a tile than the fill kernel's LDS position cache holds, and beyond 16 bits.  Same hit
    multiset as the oracle."""
    from oracle import oracle as O
    names, seqs = synth_genome(303, 400_000, 2, repeat_frac=0.05, families=2, cons_len=(300, 900))
    big, small = seqs[0].copy(), seqs[1].copy()
    big[50_000:120_000] = ord('A')
    small[1000:1040] = ord('A')      # 22 words that hit the array
    T, Q = (big, small) if big_side == 'target' else (small, big)
    g = eng.Genome(['t', 'q'], [T, Q])
    for strand in (0, 1):
        got = eng.seed_hits(g, 0, g, 1, strand)
        exp = O.seed_hits(T.tobytes(), Q.tobytes(), strand)
        assert got.size == exp.size and got.size > 1_000_000 * (strand == 0), (big_side, strand, got.size, exp.size)
        assert np.array_equal(_sorted_hits(got), _sorted_hits(exp)), (big_side, strand)
    g.close()
