"""K34's tiles of several LDS segments under independent checks (reference call site src/mimeo/wrappers.py:1025-1037).

A first-pass tile of K34 holds Lq / 4096 query entries: scaffolds of up to 5 Mbp fit one LDS segment of 1280 entries, a 10 Mbp C4
scaffold is two segments — cut at the middle key when both halves fit (a chunk then takes 12 probes or 1), else by entry count —
and a 20 Mbp super-scaffold four.  The offsets clamped to a segment, the probes a chunk may skip, the ring of pair descriptors
that runs across a segment's chunks and the prefetch wrap from a segment's last target chunk to the next segment's first only
show at those sizes.  These tests put exactly those shapes under (a) the C oracle, at sizes it finishes in seconds (its cost
goes with Lt x Lq), and (b) the round-1 decomposition of the same stage (MIMEO_HEAVY=v1: stand-alone seed scan K3 + hit array
+ K4 fast kernel) and the other forms of the first pass (MIMEO_K34_FORM), byte for byte, at C2 / C4 unit size; the packed
path's four-segment super-scaffolds are compared with the unit-per-pair path on the whole C2 job."""
import hashlib

import numpy as np
import pytest

from mimeo_amd.synth import make_families, synth_genome

pytestmark = pytest.mark.gpu

HCOLS = ['tstart', 'qstart', 'length', 'score', 'raw_score']
ACOLS = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
ENV = ('MIMEO_HEAVY', 'MIMEO_K4_VARIANT', 'MIMEO_PACK', 'MIMEO_BATCH_UNITS', 'MIMEO_K34_FORM')


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _clear(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def _two_scaffolds(seed, lt, lq, repeat_frac=0.04):
    """one scaffold of lt and one of lq bases sharing planted repeat families"""
    fams = make_families(seed, 12, (300, 4000))
    _, a = synth_genome(seed + 1, lt, 1, repeat_frac=repeat_frac, shared_families=fams)
    _, b = synth_genome(seed + 2, lq, 1, repeat_frac=repeat_frac, shared_families=fams)
    return ['t', 'q'], [a[0], b[0]]


# (target length, query length): query entries per tile = lq / 4096 -> segments of 1280; target chunks per tile =
# lt / 4096 / 64 over the eight wavefronts of a workgroup
SHAPES = [
    pytest.param(300_000, 8_000_000, id='0.3Mx8M_two_segments'),
    pytest.param(8_000_000, 300_000, id='8Mx0.3M_four_chunks_per_wavefront'),
    pytest.param(3_000_000, 12_000_000, id='3Mx12M_three_segments_prefetch_wrap'),
]


@pytest.mark.parametrize('lt,lq', SHAPES)
def test_multi_segment_hsps_and_alignments_match_the_oracle(eng, lt, lq):
    from oracle import oracle as O
    names, seqs = _two_scaffolds(lt // 1000 + lq // 100000, lt, lq)
    g = eng.Genome(names, seqs)
    T, Q = seqs[0].tobytes(), seqs[1].tobytes()
    nseg = -(-(lq // 4096) // 1280)
    for strand in (0, 1):
        got = eng.ungapped_hsps(g, 0, g, 1, strand, eng.default_params(chain=0))
        st = eng.stats()
        exp = O.ungapped_hsps(T, Q, strand, O.default_params(chain=0))
        assert exp.size > 20, (strand, exp.size)
        a, b = np.sort(got[HCOLS], order=HCOLS), np.sort(exp[HCOLS], order=HCOLS)
        assert a.size == b.size and (a == b).all(), (strand, nseg, a.size, b.size)
        # the seed-hit statistic is the sum of the pairs K34's chunk visits enumerate: an off-by-one at a segment
        # boundary shows here even when the pair it drops would have come to nothing
        assert st['seed_hits'] == O.seed_hits(T, Q, strand).size
    if lt * lq <= 3e12:   # the oracle's gapped stage on top: the whole pair
        got = eng.align_pair(g, 0, g, 1)
        exp = O.align_pair(T, Q)
        a, b = np.sort(got[ACOLS], order=ACOLS), np.sort(exp[ACOLS], order=ACOLS)
        assert a.size == b.size > 5 and (a == b).all()
    g.close()


def test_self_unit_of_two_segments_matches_the_oracle(eng):
    """a 6 Mbp scaffold against itself (1465 query entries per tile: two segments; the main diagonal goes to
    k4_diag0), plus strand: the oracle needs ~20 s"""
    from oracle import oracle as O
    names, seqs = synth_genome(606, 6_000_000, 1, repeat_frac=0.05, families=10)
    g = eng.Genome(names, seqs)
    S = seqs[0].tobytes()
    got = eng.ungapped_hsps(g, 0, g, 0, 0, eng.default_params(chain=0))
    exp = O.ungapped_hsps(S, S, 0, O.default_params(chain=0))
    a, b = np.sort(got[HCOLS], order=HCOLS), np.sort(exp[HCOLS], order=HCOLS)
    assert a.size == b.size > 50 and (a == b).all()
    assert got['length'].max() == 6_000_000
    g.close()


@pytest.mark.parametrize('seed,L', [(50, 5_000_000), (1000, 10_000_000)], ids=['c2_unit', 'c4_unit'])
def test_fullsize_units_fused_equals_round1_decomposition(eng, monkeypatch, seed, L):
    """BASELINE-sized units: K34 (default) against MIMEO_HEAVY=v1 — the stand-alone seed scan K3 writing the hit
    array and round 1's K4 fast kernel reading it: another enumeration of the hits (no segments, no frames, no
    descriptors), another filter, the same exact walks — HSPs byte for byte, both strands, a cross unit and a self
    unit, and the same number of seed hits.  And the four forms of K34's first pass (MIMEO_K34_FORM): two-segment tiles cut
    at the middle key with the level emission (a chunk of one half of the key space takes 12 probes or 1 instead of 13; a C2
    tile is one segment and must not care), tiles cut by entry count with the level emission, and either cut with the prefix
    sum + lane-major emission."""
    names, seqs = synth_genome(seed, 2 * L, 2, repeat_frac=0.05)
    g = eng.Genome(names, seqs)
    res = {}
    for tag, env in (('fused', {}), ('v1', {'MIMEO_HEAVY': 'v1'}), ('level', {'MIMEO_K34_FORM': 'level'}), ('cut', {'MIMEO_K34_FORM': 'cut'}),
                     ('half', {'MIMEO_K34_FORM': 'half'}), ('lane', {'MIMEO_K34_FORM': 'lane'})):
        _clear(monkeypatch)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out, hits = [], []
        for t, q in ((0, 1), (1, 1)):
            for strand in (0, 1):
                out.append(eng.ungapped_hsps(g, t, g, q, strand))
                hits.append(eng.stats()['seed_hits'])
        res[tag] = (out, hits)
    _clear(monkeypatch)
    assert res['fused'][1] == res['v1'][1] == res['level'][1] == res['cut'][1] == res['half'][1] == res['lane'][1]
    assert res['fused'][1][0] > 13 * float(L) * L / 4 ** 12
    for a, *others in zip(*(res[k][0] for k in ('fused', 'v1', 'level', 'cut', 'half', 'lane'))):
        assert a.size > 100 and all(a.tobytes() == o.tobytes() for o in others)
    g.close()


def test_c2_job_packed_equals_unit_per_pair_and_round1(eng, monkeypatch):
    """The whole C2 job (50 Mbp, 10 scaffolds, 200 pair-strands).  Default: the ten 5 Mbp scaffolds run as 20 Mbp
    super-scaffolds (four query segments per tile); MIMEO_PACK=0: one unit per pair (single segment); MIMEO_PACK=0
    + MIMEO_HEAVY=v1: the round-1 decomposition.  The md5 of the alignment records must be the same."""
    names, seqs = synth_genome(50, 50_000_000, 10)
    A = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(10) for q in range(10)]
    md5 = {}
    for tag, env in (('packed', {}), ('unit_per_pair', {'MIMEO_PACK': '0'}), ('round1', {'MIMEO_PACK': '0', 'MIMEO_HEAVY': 'v1'})):
        _clear(monkeypatch)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = eng.align_pairs(A, None, pairs)
        st = eng.stats()
        md5[tag] = (hashlib.md5(al.tobytes()).hexdigest(), int(al.size), st['seed_hits'])
    _clear(monkeypatch)
    assert md5['packed'][1] > 1000
    assert md5['packed'][:2] == md5['unit_per_pair'][:2] == md5['round1'][:2], md5
    assert md5['unit_per_pair'][2] == md5['round1'][2]
    A.close()
