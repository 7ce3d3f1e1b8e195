"""The C oracle (oracle/mimeo_oracle.c — what every GPU parity test holds the HIP engine to) against alignment specification v1
spelled out a second time in plain Python (tests/spec_v1.py: full matrices, dictionaries, no shared code), stage by stage, on
inputs small enough for Python loops.  Reference call site: src/mimeo/wrappers.py:1025-1037.

PARITY UNPINNED like the oracle itself: the reference holds no fixture for the LASTZ-shaped stages, so these tests show that two
restatements of the documented rules, written apart, agree — they guard the checker against its own slips (band bookkeeping,
tie-breaks, the strand mapping), they do not pin it to LASTZ."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import spec_v1 as S

BASES = np.frombuffer(b'ACGT', dtype=np.uint8)


def _rand(rng, n):
    return BASES[rng.integers(0, 4, n)].copy()


def _mutate(rng, seq, sub=0.08, indel=0.0):
    out = []
    for c in seq:
        r = rng.random()
        if r < indel / 2:
            continue                                   # deletion
        if r < indel:
            out.append(BASES[rng.integers(0, 4)])      # insertion before the base
        if rng.random() < sub:
            if rng.random() < 0.6:                     # transitions are the commoner substitution
                c = {65: 71, 71: 65, 67: 84, 84: 67}[int(c)]
            else:
                c = BASES[rng.integers(0, 4)]
        out.append(c)
    return np.array(out, dtype=np.uint8)


def _pair(seed, lt, lq, copies, cons=(120, 400), sub=0.08, indel=0.0, lower=False, ns=False, rc_copy=False):
    """a target and a query of random sequence sharing mutated copies of a few consensus sequences"""
    rng = np.random.default_rng(seed)
    T, Q = _rand(rng, lt), _rand(rng, lq)
    for k in range(copies):
        c = _rand(rng, int(rng.integers(*cons)))
        for dst, L in ((T, lt), (Q, lq)):
            m = _mutate(rng, c, sub, indel)
            if rc_copy and dst is Q and k % 2:
                m = np.frombuffer(S.revcomp(m.tobytes().decode()).encode(), dtype=np.uint8)
            p = int(rng.integers(0, L - m.size))
            dst[p:p + m.size] = m
    if ns:
        for dst, L in ((T, lt), (Q, lq)):
            for _ in range(3):
                p = int(rng.integers(0, L - 8))
                dst[p:p + int(rng.integers(1, 8))] = ord('N')
    if lower:                                          # soft-masked stretches: no seeds on the target there, scored as upper case
        for dst, L in ((T, lt), (Q, lq)):
            for _ in range(4):
                p = int(rng.integers(0, L - 60))
                seg = dst[p:p + int(rng.integers(10, 60))]
                seg[(seg >= 65) & (seg <= 90) & (seg != ord('N'))] += 32
    return T.tobytes(), Q.tobytes()


@pytest.mark.parametrize('seed,kw', [(1, {}), (2, {'lower': True, 'ns': True}), (3, {'sub': 0.02}), (4, {'rc_copy': True, 'ns': True})])
def test_seed_hits_rule_2(seed, kw):
    T, Q = _pair(seed, 1500, 1300, 5, **kw)
    for transitions in (1, 0):
        for minus in (0, 1):
            got = O.seed_hits(T, Q, minus, O.default_params(transitions=transitions))
            q = S.revcomp(Q.decode()) if minus else Q.decode()
            exp = S.seed_hits(T.decode(), q, bool(transitions))
            assert [(int(a), int(b)) for a, b in zip(got['tpos'], got['qpos'])] == exp, (seed, transitions, minus)
            if not minus and transitions:
                assert len(exp) > 50


@pytest.mark.parametrize('seed,kw', [(11, {}), (12, {'lower': True, 'ns': True}), (13, {'sub': 0.15}), (14, {'rc_copy': True})])
def test_gap_free_extension_threshold_and_entropy_rules_3_and_4(seed, kw):
    T, Q = _pair(seed, 2500, 2200, 6, **kw)
    T2 = T[:1200] + b'AT' * 90 + T[1200:]              # a low-complexity stretch on both: the entropy factor bites
    Q2 = Q[:700] + b'AT' * 80 + Q[700:]
    for (t, q) in ((T, Q), (T2, Q2)):
        for params in ({}, {'entropy': 0}, {'hspthresh': 2200, 'xdrop': 340}, {'transitions': 0}):
            for minus in (0, 1):
                got = O.ungapped_hsps(t, q, minus, O.default_params(chain=0, **params))
                qq = S.revcomp(q.decode()) if minus else q.decode()
                exp = S.ungapped_hsps(t.decode(), qq, params.get('hspthresh', 3000), params.get('xdrop', 910),
                                      bool(params.get('transitions', 1)), bool(params.get('entropy', 1)))
                g = sorted((int(h['tstart']), int(h['qstart']), int(h['length']), int(h['score']), int(h['raw_score'])) for h in got)
                assert g == sorted(exp), (seed, params, minus)
    assert len(exp) >= 0


@pytest.mark.parametrize('seed,kw,params', [
    (21, {'indel': 0.02}, {}),
    (22, {'indel': 0.03, 'ns': True, 'lower': True}, {}),
    (23, {'indel': 0.02, 'rc_copy': True}, {'ydrop': 3000}),
    (24, {'indel': 0.04, 'sub': 0.12}, {'gap_open': 200, 'gap_extend': 60, 'hspthresh': 2500}),
    # five copies, every other one reverse-complemented in the query: both strands align, and the chained HSPs outnumber the
    # alignments (their anchors fall inside the box of the first alignment of the strand and are skipped)
    (30, {'indel': 0.02, 'rc_copy': True, 'big': True}, {}),
    (36, {'indel': 0.02, 'rc_copy': True, 'big': True}, {}),
])
def test_whole_pair_rules_5_to_7(seed, kw, params):
    """chain, anchors (box rule), two one-sided y-drop DPs per anchor, the gapped threshold, minus-strand coordinates:
    orc_align_pair against the full-matrix restatement"""
    kw = dict(kw)
    big = kw.pop('big', False)
    T, Q = _pair(seed, 900, 850, 5, cons=(140, 260), **kw) if big else _pair(seed, 700, 650, 3, cons=(150, 330), **kw)
    got = O.align_pair(T, Q, O.default_params(**params))
    exp = []
    for minus in (0, 1):
        exp += S.align_strand(T.decode(), Q.decode(), minus, params.get('hspthresh', 3000), 910, params.get('ydrop', 9400),
                              params.get('gap_open', 400), params.get('gap_extend', 30))
    g = sorted((int(a['tstart']), int(a['tend']), int(a['qstart']), int(a['qend']), int(a['score']), int(a['id_n']), int(a['id_d']),
                int(a['qstrand'])) for a in got)
    assert g == sorted(exp), (seed, g, exp)
    assert len(exp) >= 1 and (not big or {e[7] for e in exp} == {0, 1})
