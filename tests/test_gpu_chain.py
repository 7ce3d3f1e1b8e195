"""K5 alone — lastz `--chain` (reference call site src/mimeo/wrappers.py:1031) — on HSP sets made for it, through
`mimeo_chain_hsps`, against the oracle's O(n^2) chain (`oracle/mimeo_oracle.c: chain_hsps`): flags must be identical.

Three kernels share the recurrence and its tie rule (earliest predecessor, earliest end): `k5_chain` (tiles of 64, groups up to
32 768 HSPs), `k5_chain_wave` (sweep with a Fenwick tree over query ends: larger groups) and `k5_chain_big` (blocks of 2048: groups
of 2^24 HSPs and more).  MIMEO_K5_BIG_MIN / MIMEO_K5_BIG / MIMEO_K5_NO_BIG send the same HSPs through each of them.  The sets:
random boxes, collinear runs (every HSP the next one's predecessor: nothing but inner dependencies), microsatellite-like
rectangles (hundreds of HSPs that start or end on one base: waves and wide windows), all scores equal (ties everywhere), and sizes
around the tile (64), the tree block (1024) and the window share boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _unique(h):
    key = np.stack([h['tstart'], h['qstart'], h['length']], 1)
    _, idx = np.unique(key, axis=0, return_index=True)
    return h[np.sort(idx)]


def _make(kind, n, rng):
    from mimeo_amd import _ffi
    h = np.zeros(n, dtype=_ffi.HSP)
    if kind == 'random':
        L = max(1000, n * 20)
        h['tstart'] = rng.integers(0, L, n)
        h['qstart'] = rng.integers(0, L, n)
        h['length'] = rng.integers(30, 400, n)
        h['score'] = h['length'].astype(np.int64) * 90 - rng.integers(0, 2000, n)
    elif kind == 'collinear':
        # pieces along a few diagonals, each piece starting where the one before ended (or a little later / earlier)
        step = rng.integers(20, 80, n)
        jit = rng.integers(-3, 12, n)
        t = np.cumsum(step + jit)
        d = rng.integers(0, 4, n) * 1000
        h['tstart'] = t + 5000
        h['qstart'] = t + 5000 - d + rng.integers(0, 3, n)
        h['length'] = step
        h['score'] = step.astype(np.int64) * 95 - rng.integers(0, 300, n)
    elif kind == 'rectangles':
        # arrays of one motif: every in-phase diagonal of (target array, query array) is an HSP; many start on the array's
        # first base, many end on its last
        out = []
        na = max(2, int(np.sqrt(n / 40)))
        ta = np.sort(rng.integers(0, 200 * na, na)) * 5
        qa = np.sort(rng.integers(0, 200 * na, na)) * 5
        tl, ql = rng.integers(60, 400, na), rng.integers(60, 400, na)
        for i in range(na):
            for j in range(na):
                p = int(rng.integers(1, 7))
                for dd in range(-int(ql[j]) + 30, int(tl[i]) - 30, p):
                    ts = ta[i] + max(0, dd)
                    qs = qa[j] + max(0, -dd)
                    ln = min(ta[i] + tl[i] - ts, qa[j] + ql[j] - qs)
                    if ln >= 30:
                        out.append((ts, qs, ln))
        out = np.array(out[:n] if len(out) >= n else out, dtype=np.int64)
        h = np.zeros(out.shape[0], dtype=_ffi.HSP)
        h['tstart'], h['qstart'], h['length'] = out[:, 0], out[:, 1], out[:, 2]
        h['score'] = h['length'].astype(np.int64) * 91
    elif kind == 'ties':
        L = max(500, n * 4)
        h['tstart'] = rng.integers(0, L, n)
        h['qstart'] = rng.integers(0, L, n)
        h['length'] = 40
        h['score'] = 3000
    h['raw_score'] = h['score'] + 7
    h = _unique(h)
    return h[rng.permutation(h.size)]   # the engine sorts


def _check(eng, O, h, tag):
    got = eng.chain_hsps(h)
    exp = O.chain_hsps(h)
    assert got.size == exp.size == h.size
    for f in ('tstart', 'qstart', 'length', 'score', 'raw_score'):
        assert np.array_equal(got[f], exp[f]), (tag, f)
    bad = np.flatnonzero((got['flags'] & 1) != (exp['flags'] & 1))
    assert bad.size == 0, (tag, h.size, bad[:8], got[bad[:8]], exp[bad[:8]])
    return int((exp['flags'] & 1).sum())


MODES = {'wave': {'MIMEO_K5_BIG_MIN': '1'}, 'two-level': {'MIMEO_K5_BIG_MIN': '1', 'MIMEO_K5_BIG': 'old'}, 'tiles': {'MIMEO_K5_NO_BIG': '1'}}


@pytest.mark.parametrize('mode', list(MODES))
def test_chain_kernels_on_made_up_hsps(eng, monkeypatch, mode):
    from oracle import oracle as O
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(20260 + len(mode))
    chained = 0
    for kind in ('random', 'collinear', 'rectangles', 'ties'):
        for n in (1, 2, 3, 63, 64, 65, 127, 129, 511, 1023, 1024, 1025, 2047, 2049, 3000, 4097, 9000):
            h = _make(kind, n, rng)
            if h.size:
                chained += _check(eng, O, h, (mode, kind, n))
    assert chained > 1000


def test_chain_default_dispatch_on_a_large_group(eng, monkeypatch):
    """beyond 32 768 HSPs the wave kernel is the default: 40 000 of each kind, nothing set"""
    from oracle import oracle as O
    for k in ('MIMEO_K5_BIG_MIN', 'MIMEO_K5_BIG', 'MIMEO_K5_NO_BIG'):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(7)
    for kind in ('random', 'collinear', 'rectangles', 'ties'):
        h = _make(kind, 40_000, rng)
        assert h.size > 33_000, (kind, h.size)
        assert _check(eng, O, h, (kind, 'large')) > 0


def test_wave_and_two_level_kernels_agree_on_a_million_hsps(eng, monkeypatch):
    """beyond what the O(n^2) oracle can check: 1.2 * 10^6 HSPs (rectangles of a microsatellite-rich pair mixed with random
    boxes and collinear runs) through the wave kernel and through the two-level kernel — two algorithms, one set of flags"""
    for k in ('MIMEO_K5_BIG_MIN', 'MIMEO_K5_BIG', 'MIMEO_K5_NO_BIG'):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(99)
    parts = [_make('rectangles', 600_000, rng), _make('random', 400_000, rng), _make('collinear', 200_000, rng)]
    parts[1]['tstart'] += 1000; parts[2]['qstart'] += 500
    h = _unique(np.concatenate(parts))
    assert h.size > 1_000_000
    wave = eng.chain_hsps(h)
    monkeypatch.setenv('MIMEO_K5_BIG', 'old')
    old = eng.chain_hsps(h)
    monkeypatch.delenv('MIMEO_K5_BIG')
    assert wave.tobytes() == old.tobytes()
    assert int((wave['flags'] & 1).sum()) > 100
