"""Runs of consecutive seed hits (k4_device.h "runs of consecutive seed hits"; reference call site src/mimeo/wrappers.py:1025-1037,
`--gfextend` with lastz's per-diagonal "already extended" rule).

Two microsatellites of one motif are a rectangle of seed hits: every in-phase diagonal carries hits at consecutive positions, of
which all but the first are followers of their left neighbour.  The engine records such a run by its two ends (the walk kernel
classifies a member by three exact seed tests; K34's split pass drops interior members before its pre-filter) and the resolution
kernels read a RUN_END record as the range it closes.  Whatever the representation, the HSPs must be the sequential oracle's —
with N runs and soft-masked stretches inside and beside the arrays, at the ends of the scaffolds, on both strands, and with the
split-pass shortcut switched off (MIMEO_K34_DEBUG=16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HCOLS = ['tstart', 'qstart', 'length', 'score', 'raw_score']
ACGT = np.frombuffer(b'ACGT', np.uint8)


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _arrays(rng, n, motifs, copies, noise):
    """a random scaffold of n bases with `copies` arrays (50-500 bp) of the given motifs; `noise`: substitution rate inside them"""
    s = rng.integers(0, 4, size=n, dtype=np.uint8)
    where = []
    for _ in range(copies):
        m = motifs[int(rng.integers(0, len(motifs)))]
        ln = int(rng.integers(50, 501))
        p = int(rng.integers(0, n - ln))
        a = np.resize(np.array(m, dtype=np.uint8), ln)
        if noise:
            k = rng.random(ln) < noise
            a[k] = (a[k] + rng.integers(1, 4, size=int(k.sum()), dtype=np.uint8)) & 3
        s[p:p + ln] = a
        where.append((p, ln))
    return s, where


def _check(eng, O, names, seqs, pairs, **kw):
    g = eng.Genome(names, seqs)
    nh, stats = 0, []
    for t, q in pairs:
        for strand in (0, 1):
            got = eng.ungapped_hsps(g, t, g, q, strand, eng.default_params(chain=0, **kw))
            stats.append(eng.stats())
            exp = O.ungapped_hsps(seqs[t].tobytes(), seqs[q].tobytes(), strand, O.default_params(chain=0, **kw))
            a, b = np.sort(got[HCOLS], order=HCOLS), np.sort(exp[HCOLS], order=HCOLS)
            assert a.size == b.size and (a == b).all(), (t, q, strand, a.size, b.size)
            nh += b.size
    g.close()
    return nh, stats


@pytest.mark.parametrize('noise', [0.0, 0.03])
def test_microsatellite_rectangles_match_the_oracle(eng, monkeypatch, noise):
    from oracle import oracle as O
    rng = np.random.default_rng(11 + int(noise * 100))
    motifs = [[0], [3], [1, 0], [2, 3], [0, 1, 2], [3, 3, 0], [0, 1, 2, 3], [2, 0, 0, 1, 3], [1, 1, 0, 2, 3, 3]]
    seqs, where = [], []
    for _ in range(3):
        s, w = _arrays(rng, 150_000, motifs, 60, noise)
        seqs.append(ACGT[s])
        where.append(w)
    # N inside an array, one base beside one, a run of N through one; a soft-masked array and a soft-masked half of one (target role only)
    s0, s1 = seqs[0].copy(), seqs[1].copy()
    for k, (p, ln) in enumerate(where[0][:12]):
        if k % 4 == 0: s0[p + ln // 2] = ord('N')
        if k % 4 == 1: s0[max(0, p - 1)] = ord('N')
        if k % 4 == 2: s0[p + 10:p + 10 + 25] = ord('N')
        if k % 4 == 3: s0[p:p + ln] |= 0x20
    for k, (p, ln) in enumerate(where[1][:8]):
        if k % 2 == 0: s1[p + ln // 3:p + ln] |= 0x20
        else: s1[p + ln - 1] = ord('N')
    # arrays flush with both ends of a scaffold
    s2 = seqs[2].copy()
    s2[:300] = ACGT[np.resize(np.array([0, 1], dtype=np.uint8), 300)]
    s2[-200:] = ACGT[np.resize(np.array([0], dtype=np.uint8), 200)]
    seqs = [s0, s1, s2]
    names = ['m0', 'm1', 'm2']
    pairs = [(0, 1), (1, 0), (2, 0), (1, 2), (2, 2)]
    monkeypatch.delenv('MIMEO_K34_DEBUG', raising=False)
    nh, st = _check(eng, O, names, seqs, pairs)
    assert nh > 200
    hits = sum(s['seed_hits'] for s in st)
    fol = sum(s['followers'] for s in st)
    # a rectangle's hits are nearly all followers; an unbroken run leaves two records (3 % substitutions cut the runs into short pieces)
    assert fol < (0.05 if noise == 0.0 else 0.6) * hits, (fol, hits)
    # the split-pass shortcut off: every member goes through the walk kernel, same HSPs, same records
    monkeypatch.setenv('MIMEO_K34_DEBUG', '16')
    nh2, st2 = _check(eng, O, names, seqs, pairs)
    monkeypatch.delenv('MIMEO_K34_DEBUG')
    assert nh2 == nh and sum(s['followers'] for s in st2) == fol
    assert sum(s['walked_hits'] for s in st2) >= sum(s['walked_hits'] for s in st)


def test_runs_inside_diverged_repeat_copies_and_low_thresholds(eng):
    """real similarity: runs of a few consecutive hits broken by mismatches, extensions that reach into the next run or stop
    short of it (low x-drop), thresholds low enough that many of the pieces are HSPs of their own"""
    from oracle import oracle as O
    from mimeo_amd.synth import synth_genome
    for seed, kw in ((5, dict(xdrop=500, hspthresh=1500)), (6, dict(xdrop=910, hspthresh=3000, transitions=0)), (7, dict(xdrop=640, hspthresh=2200, entropy=0))):
        names, seqs = synth_genome(seed, 400_000, 2, repeat_frac=0.5, families=3, cons_len=(200, 1500), max_div=0.3, indel_rate=0.02, microsat_frac=0.01)
        nh, _ = _check(eng, O, names, seqs, [(0, 1), (0, 0)], **kw)
        assert nh > 100
